// api_interop.hip — C ABI of the instance pipeline, part 4 of 4: zero-copy interop with the renderer's own Vulkan
// allocations and queues: memory exported as a POSIX fd (dma-buf on amdgpu) mapped into the HIP
// device, and timeline / binary semaphores exported as fds waited for and signalled in stream order.
#include "context.hpp"

#include <drm/drm.h>  // DRM sync objects: what an exported Vulkan semaphore fd is on amdgpu (kernel uapi, no libdrm)
#include <fcntl.h>
#include <sys/ioctl.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>

namespace mip_host {

void interop_release(MipContext* ctx) {
  for (auto& e : ctx->externals) (void)hipDestroyExternalMemory(e.mem);  // unmaps the buffer as well
  ctx->externals.clear();
  for (auto* e : ctx->semaphores) {
    if (e->sem) (void)hipDestroyExternalSemaphore(e->sem);
    if (e->drm_handle && ctx->drm_fd >= 0) {
      drm_syncobj_destroy d{};
      d.handle = e->drm_handle;
      (void)ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_DESTROY, &d);
    }
    delete e;
  }
  ctx->semaphores.clear();
  if (ctx->drm_fd >= 0) close(ctx->drm_fd);
  ctx->drm_fd = -1;
}

}  // namespace mip_host

using namespace mip_host;

extern "C" {

int32_t mip_import_external_fd(MipContext* ctx, int32_t fd, uint64_t size_bytes, void** out_device_ptr) {
  if (out_device_ptr) *out_device_ptr = nullptr;
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (fd < 0 || size_bytes == 0 || !out_device_ptr) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad fd / size / out pointer");
  if (int32_t rc = bind_device(ctx)) return rc;
  hipExternalMemoryHandleDesc hd{};
  hd.type = hipExternalMemoryHandleTypeOpaqueFd;
  hd.handle.fd = fd;
  hd.size = size_bytes;
  hipExternalMemory_t mem = nullptr;
  hipError_t e = hipImportExternalMemory(&mem, &hd);
  if (e != hipSuccess) return fail(ctx, MIP_ERR_DEVICE, "hipImportExternalMemory(OpaqueFd, %llu bytes) failed: %s", (unsigned long long)size_bytes, hipGetErrorString(e));
  hipExternalMemoryBufferDesc bd{};
  bd.offset = 0;
  bd.size = size_bytes;
  void* ptr = nullptr;
  e = hipExternalMemoryGetMappedBuffer(&ptr, mem, &bd);
  if (e != hipSuccess || !ptr) {
    (void)hipDestroyExternalMemory(mem);
    return fail(ctx, MIP_ERR_DEVICE, "hipExternalMemoryGetMappedBuffer failed: %s", hipGetErrorString(e));
  }
  ctx->externals.push_back({mem, ptr});
  *out_device_ptr = ptr;
  return MIP_OK;
}

int32_t mip_release_external(MipContext* ctx, void* device_ptr) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < ctx->externals.size(); ++i)
    if (ctx->externals[i].ptr == device_ptr) {
      if (int32_t rc = bind_device(ctx)) return rc;
      if (int32_t rc = sync_all(ctx)) return rc;
      MIP_HIP(ctx, hipDestroyExternalMemory(ctx->externals[i].mem));
      ctx->externals.erase(ctx->externals.begin() + (long)i);
      return MIP_OK;
    }
  return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a pointer returned by mip_import_external_fd");
}

static MipContext::ExternalSemaphore* find_semaphore(MipContext* ctx, MipExternalSemaphore* h, size_t* at = nullptr) {
  for (size_t i = 0; i < ctx->semaphores.size(); ++i)
    if ((void*)ctx->semaphores[i] == (void*)h) {
      if (at) *at = i;
      return ctx->semaphores[i];
    }
  return nullptr;
}

// ---- DRM sync object path (host functions on the stream) ----
struct SemaphoreOp {
  int drm_fd;
  uint32_t handle, kind;
  uint64_t value;
  bool signal;
  volatile uint32_t* error_word;  // host memory (the context's error words): a wait that expired is reported by the next mip_wait
};
constexpr int64_t kSemaphoreWaitNs = 10ll * 1000 * 1000 * 1000;  // bounded like every other wait in the library

// ioctl restarted when a signal interrupts it (what libdrm's drmIoctl does; the waits carry an ABSOLUTE deadline)
static int drm_ioctl(int fd, unsigned long request, void* arg) {
  int rc;
  do rc = ioctl(fd, request, arg);
  while (rc == -1 && (errno == EINTR || errno == EAGAIN));
  return rc;
}

static void semaphore_host_fn(void* p) {
  SemaphoreOp* op = static_cast<SemaphoreOp*>(p);
  uint32_t handle = op->handle;
  uint64_t point = op->value;
  int rc = 0;
  if (op->signal) {
    if (op->kind == MIP_SEMAPHORE_TIMELINE) {
      drm_syncobj_timeline_array a{};
      a.handles = (uintptr_t)&handle;
      a.points = (uintptr_t)&point;
      a.count_handles = 1;
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_TIMELINE_SIGNAL, &a);
    } else {
      drm_syncobj_array a{};
      a.handles = (uintptr_t)&handle;
      a.count_handles = 1;
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_SIGNAL, &a);
    }
  } else {
    timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    const int64_t deadline = (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec + kSemaphoreWaitNs;
    if (op->kind == MIP_SEMAPHORE_TIMELINE) {
      drm_syncobj_timeline_wait w{};
      w.handles = (uintptr_t)&handle;
      w.points = (uintptr_t)&point;
      w.timeout_nsec = deadline;
      w.count_handles = 1;
      w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;  // the point may not have been submitted yet
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_TIMELINE_WAIT, &w);
    } else {
      drm_syncobj_wait w{};
      w.handles = (uintptr_t)&handle;
      w.timeout_nsec = deadline;
      w.count_handles = 1;
      w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_WAIT, &w);
      if (rc == 0) {  // a binary semaphore is consumed by its wait
        drm_syncobj_array a{};
        a.handles = (uintptr_t)&handle;
        a.count_handles = 1;
        (void)drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_RESET, &a);
      }
    }
  }
  if (rc != 0) *op->error_word = mip::kErrSemaphore;
  delete op;
}

static int32_t enqueue_semaphore_op(MipContext* ctx, MipContext::ExternalSemaphore* s, uint64_t value, bool signal, hipStream_t stream) {
  SemaphoreOp* op = new (std::nothrow) SemaphoreOp{ctx->drm_fd, s->drm_handle, s->kind, value, signal, ctx->h_error + 5};
  if (!op) return fail(ctx, MIP_ERR_OUT_OF_MEMORY, "out of host memory");
  const hipError_t e = hipLaunchHostFunc(stream, semaphore_host_fn, op);
  if (e != hipSuccess) {
    delete op;
    return fail(ctx, MIP_ERR_DEVICE, "hipLaunchHostFunc failed: %s", hipGetErrorString(e));
  }
  return MIP_OK;
}

int32_t mip_import_external_semaphore_fd(MipContext* ctx, int32_t fd, uint32_t kind, MipExternalSemaphore** out_semaphore) {
  if (out_semaphore) *out_semaphore = nullptr;
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (fd < 0 || !out_semaphore) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad fd / out pointer");
  if (kind != MIP_SEMAPHORE_BINARY && kind != MIP_SEMAPHORE_TIMELINE) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "kind %u is neither MIP_SEMAPHORE_BINARY nor MIP_SEMAPHORE_TIMELINE", kind);
  if (int32_t rc = bind_device(ctx)) return rc;
  hipExternalSemaphoreHandleDesc hd{};
  hd.type = kind == MIP_SEMAPHORE_TIMELINE ? hipExternalSemaphoreHandleTypeTimelineSemaphoreFd : hipExternalSemaphoreHandleTypeOpaqueFd;
  hd.handle.fd = fd;
  hipExternalSemaphore_t sem = nullptr;
  hipError_t e = std::getenv("MIP_TUNE_SEMAPHORE_VIA_DRM") ? hipErrorNotSupported : hipImportExternalSemaphore(&sem, &hd);
  uint32_t drm_handle = 0;
  if (e != hipSuccess || !sem) {
    // The runtime refuses the handle type (ROCm 7.2, Linux: TimelineSemaphoreFd -> "invalid argument", OpaqueFd ->
    // "operation not supported"). The fd itself is a kernel sync object: take it on a render node.
    sem = nullptr;
    (void)hipGetLastError();
    if (ctx->drm_fd < 0) {
      char node[64];
      for (int k = 128; k < 192 && ctx->drm_fd < 0; ++k) {
        snprintf(node, sizeof node, "/dev/dri/renderD%d", k);
        ctx->drm_fd = open(node, O_RDWR | O_CLOEXEC);
      }
    }
    drm_syncobj_handle h{};
    h.fd = fd;
    if (ctx->drm_fd < 0 || drm_ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_FD_TO_HANDLE, &h) != 0 || !h.handle)
      return fail(ctx, MIP_ERR_DEVICE, "hipImportExternalSemaphore(%s) failed: %s; and the fd is not a DRM sync object either (%s)",
                  kind == MIP_SEMAPHORE_TIMELINE ? "TimelineSemaphoreFd" : "OpaqueFd", hipGetErrorString(e),
                  ctx->drm_fd < 0 ? "no render node could be opened" : "DRM_IOCTL_SYNCOBJ_FD_TO_HANDLE refused it");
    drm_handle = h.handle;
    close(fd);  // imported: the fd belonged to the library from here on (the sync object lives on through the handle)
  }
  auto* entry = new (std::nothrow) MipContext::ExternalSemaphore{sem, kind, drm_handle};
  if (!entry) {
    if (sem) (void)hipDestroyExternalSemaphore(sem);
    return fail(ctx, MIP_ERR_OUT_OF_MEMORY, "out of host memory");
  }
  ctx->semaphores.push_back(entry);
  *out_semaphore = (MipExternalSemaphore*)entry;
  return MIP_OK;
}

int32_t mip_external_semaphore_on_device(MipContext* ctx, MipExternalSemaphore* semaphore) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  return s->sem ? 1 : 0;
}

int32_t mip_wait_external(MipContext* ctx, MipExternalSemaphore* semaphore, uint64_t value) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  if (int32_t rc = bind_device(ctx)) return rc;
  // the stream the NEXT frame will be enqueued on: that frame then starts only when the semaphore has been reached
  hipStream_t stream = ctx->slots[ctx->next_slot].stream;
  if (s->sem) {
    hipExternalSemaphoreWaitParams p{};
    p.params.fence.value = value;
    MIP_HIP(ctx, hipWaitExternalSemaphoresAsync(&s->sem, &p, 1, stream));
  } else if (int32_t rc = enqueue_semaphore_op(ctx, s, value, false, stream)) {
    return rc;
  }
  ctx->pending_async = true;
  return MIP_OK;
}

int32_t mip_signal_external(MipContext* ctx, MipExternalSemaphore* semaphore, uint64_t value) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  if (int32_t rc = bind_device(ctx)) return rc;
  // behind the frame that was issued last (its slot's stream)
  hipStream_t stream = ctx->slots[ctx->last_slot].stream;
  if (s->sem) {
    hipExternalSemaphoreSignalParams p{};
    p.params.fence.value = value;
    MIP_HIP(ctx, hipSignalExternalSemaphoresAsync(&s->sem, &p, 1, stream));
  } else if (int32_t rc = enqueue_semaphore_op(ctx, s, value, true, stream)) {
    return rc;
  }
  ctx->pending_async = true;
  return MIP_OK;
}

int32_t mip_release_external_semaphore(MipContext* ctx, MipExternalSemaphore* semaphore) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  size_t at = 0;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore, &at);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (s->sem) MIP_HIP(ctx, hipDestroyExternalSemaphore(s->sem));
  if (s->drm_handle) {
    drm_syncobj_destroy d{};
    d.handle = s->drm_handle;
    (void)ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_DESTROY, &d);
  }
  ctx->semaphores.erase(ctx->semaphores.begin() + (long)at);
  delete s;
  return MIP_OK;
}

}  // extern "C"
