// api_interop.hip — C ABI of the instance pipeline, part 4 of 4: zero-copy interop with the renderer's own Vulkan
// allocations and queues: memory exported as a POSIX fd (dma-buf on amdgpu) mapped into the HIP
// device, and timeline / binary semaphores exported as fds waited for and signalled in stream order.
#include <atomic>

#include "context.hpp"

#include <drm/drm.h>  // DRM sync objects: what an exported Vulkan semaphore fd is on amdgpu (kernel uapi, no libdrm)
#include <fcntl.h>
#include <sys/ioctl.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace mip_host {

void stop_semaphore_workers(MipContext* ctx);

void interop_release(MipContext* ctx) {
  stop_semaphore_workers(ctx);
  for (auto& e : ctx->externals) (void)hipDestroyExternalMemory(e.mem);  // unmaps the buffer as well
  ctx->externals.clear();
  for (auto* e : ctx->semaphores) {
    if (e->sem) (void)hipDestroyExternalSemaphore(e->sem);
    if (e->drm_handle && ctx->drm_fd >= 0) {
      drm_syncobj_destroy d{};
      d.handle = e->drm_handle;
      (void)ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_DESTROY, &d);
    }
    if (e->words) (void)hipHostFree(e->words);
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

// One wait or signal on the kernel object; false when the ioctl failed or the 10 s expired.
static bool perform_semaphore_op(int drm_fd, uint32_t handle_in, uint32_t kind, uint64_t point_in, bool signal) {
  uint32_t handle = handle_in;
  uint64_t point = point_in;
  int rc = 0;
  if (signal) {
    if (kind == MIP_SEMAPHORE_TIMELINE) {
      drm_syncobj_timeline_array a{};
      a.handles = (uintptr_t)&handle;
      a.points = (uintptr_t)&point;
      a.count_handles = 1;
      rc = drm_ioctl(drm_fd, DRM_IOCTL_SYNCOBJ_TIMELINE_SIGNAL, &a);
    } else {
      drm_syncobj_array a{};
      a.handles = (uintptr_t)&handle;
      a.count_handles = 1;
      rc = drm_ioctl(drm_fd, DRM_IOCTL_SYNCOBJ_SIGNAL, &a);
    }
  } else {
    timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    const int64_t deadline = (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec + kSemaphoreWaitNs;
    if (kind == MIP_SEMAPHORE_TIMELINE) {
      drm_syncobj_timeline_wait w{};
      w.handles = (uintptr_t)&handle;
      w.points = (uintptr_t)&point;
      w.timeout_nsec = deadline;
      w.count_handles = 1;
      w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;  // the point may not have been submitted yet
      rc = drm_ioctl(drm_fd, DRM_IOCTL_SYNCOBJ_TIMELINE_WAIT, &w);
    } else {
      drm_syncobj_wait w{};
      w.handles = (uintptr_t)&handle;
      w.timeout_nsec = deadline;
      w.count_handles = 1;
      w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;
      rc = drm_ioctl(drm_fd, DRM_IOCTL_SYNCOBJ_WAIT, &w);
      if (rc == 0) {  // a binary semaphore is consumed by its wait
        drm_syncobj_array a{};
        a.handles = (uintptr_t)&handle;
        a.count_handles = 1;
        (void)drm_ioctl(drm_fd, DRM_IOCTL_SYNCOBJ_RESET, &a);
      }
    }
  }
  return rc == 0;
}

// (fall-back path: runtimes without hipStreamWaitValue64, or MIP_TUNE_SEMAPHORE_HOST_FUNCTIONS=1 for an A/B)
static void semaphore_host_fn(void* p) {
  SemaphoreOp* op = static_cast<SemaphoreOp*>(p);
  if (!perform_semaphore_op(op->drm_fd, op->handle, op->kind, op->value, op->signal)) *op->error_word = mip::kErrSemaphore;
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

}  // extern "C"

// ---- DRM sync object path WITHOUT host functions on the stream -----------------------------------------------------------
// Measured (tools/micro/semaphore_frames.cpp, profiles/r04_external_semaphore_handover.txt): with the wait and the signal of a
// frame as two hipLaunchHostFunc calls, a 1 M-instance frame that takes 17.7 us bare takes 110 us (80 us in a ping-pong
// with a consumer thread): ~45 us per host function. Instead:
//   wait    the stream holds a hipStreamWaitValue64 on a pinned word (the hardware waits, no runtime thread involved); a
//           WAITER thread of the library blocks in the DRM wait ioctl and stores the sequence number the stream is waiting for;
//   signal  the stream stores a sequence number into a second pinned word (hipStreamWriteValue64) behind the frame; a
//           SIGNALLER thread polls that word and performs the DRM signal ioctl.
// Two threads, because a wait that blocks (the consumer of frame k has not finished) must not keep the signal of frame k
// from going out — the consumer is waiting for exactly that. The threads sleep on a condition variable while nothing is
// queued; the signaller spins only while a frame whose signal is queued is on the device.
// words[0]: the wait word (granted sequence number); words[kSignalWord0 + seq % kSignalRing]: the word signal request `seq` is
// written to by its stream (a cache line away from the wait word)
static constexpr unsigned kSignalWord0 = 8, kSignalRing = 32, kSemaphoreWordBytes = (kSignalWord0 + kSignalRing) * 8;

struct MipContext::SemaphoreWorkers {
  struct Request {
    MipContext::ExternalSemaphore* sem;
    uint64_t value;
    unsigned long long seq;
  };
  std::mutex m;
  std::condition_variable wake_waiter, wake_signaller, idle;
  std::deque<Request> waits, signals;
  bool waiter_busy = false, signaller_busy = false;
  std::atomic<bool> stop{false};  // set under m (the condition variables read it there); the signaller's spin reads it without
  std::thread waiter, signaller;
  int drm_fd = -1;
  volatile uint32_t* error_word = nullptr;
};

namespace mip_host {

static void waiter_main(MipContext::SemaphoreWorkers* w) {
  std::unique_lock<std::mutex> lk(w->m);
  for (;;) {
    w->wake_waiter.wait(lk, [&] { return w->stop || !w->waits.empty(); });
    if (w->stop) return;
    const auto rq = w->waits.front();
    w->waits.pop_front();
    w->waiter_busy = true;
    lk.unlock();
    if (!perform_semaphore_op(w->drm_fd, rq.sem->drm_handle, rq.sem->kind, rq.value, false)) *w->error_word = mip::kErrSemaphore;
    // granted (or expired: reported, and the frame behind it runs anyway, as the header says): release the stream
    __atomic_store_n(&rq.sem->words[0], rq.seq, __ATOMIC_RELEASE);
    lk.lock();
    w->waiter_busy = false;
    w->idle.notify_all();
  }
}

static void signaller_main(MipContext::SemaphoreWorkers* w) {
  std::unique_lock<std::mutex> lk(w->m);
  for (;;) {
    w->wake_signaller.wait(lk, [&] { return w->stop || !w->signals.empty(); });
    if (w->stop) return;
    const auto rq = w->signals.front();
    w->signals.pop_front();
    w->signaller_busy = true;
    lk.unlock();
    // the stream stores rq.seq when everything in front of the signal has finished
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    uint32_t polls = 0;
    bool reached = true;
    // (this request's own ring word; entries are reused every kSignalRing requests with larger numbers, never smaller)
    while (__atomic_load_n(&rq.sem->words[kSignalWord0 + rq.seq % kSignalRing], __ATOMIC_ACQUIRE) != rq.seq) {
      if (w->stop.load(std::memory_order_relaxed)) { reached = false; break; }
      if (++polls < 4096u) continue;  // a frame is tens of microseconds: spin that long ...
      timespec now;
      clock_gettime(CLOCK_MONOTONIC, &now);
      if ((now.tv_sec - t0.tv_sec) * 1000000000ll + (now.tv_nsec - t0.tv_nsec) > 2 * kSemaphoreWaitNs) { reached = false; break; }
      timespec nap{0, 20000};         // ... then look every 20 us
      nanosleep(&nap, nullptr);
    }
    if (!reached || !perform_semaphore_op(w->drm_fd, rq.sem->drm_handle, rq.sem->kind, rq.value, true)) *w->error_word = mip::kErrSemaphore;
    lk.lock();
    w->signaller_busy = false;
    w->idle.notify_all();
  }
}

void stop_semaphore_workers(MipContext* ctx) {
  MipContext::SemaphoreWorkers* w = ctx->semaphore_workers;
  if (!w) return;
  {
    std::lock_guard<std::mutex> lk(w->m);
    w->stop = true;
  }
  w->wake_waiter.notify_all();
  w->wake_signaller.notify_all();
  if (w->waiter.joinable()) w->waiter.join();
  if (w->signaller.joinable()) w->signaller.join();
  delete w;
  ctx->semaphore_workers = nullptr;
}

// Every queued wait has been granted and every queued signal performed (the streams have drained: mip_wait).
int32_t interop_drain(MipContext* ctx) {
  MipContext::SemaphoreWorkers* w = ctx->semaphore_workers;
  if (!w) return MIP_OK;
  std::unique_lock<std::mutex> lk(w->m);
  const bool ok = w->idle.wait_for(lk, std::chrono::seconds(30), [&] { return w->waits.empty() && w->signals.empty() && !w->waiter_busy && !w->signaller_busy; });
  return ok ? MIP_OK : fail(ctx, MIP_ERR_TIMEOUT, "the external-semaphore helper threads did not drain");
}

}  // namespace mip_host

static bool stream_values_usable(MipContext* ctx) {
  static int usable = -1;
  if (usable < 0) {
    int can = 0;
    usable = (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) == hipSuccess && can) ? 1 : 0;
    (void)hipGetLastError();
    if (std::getenv("MIP_TUNE_SEMAPHORE_HOST_FUNCTIONS")) usable = 0;  // A/B: the host-function path of round 3
  }
  return usable == 1;
}

// The DRM path of mip_wait_external / mip_signal_external.
static int32_t enqueue_drm_semaphore(MipContext* ctx, MipContext::ExternalSemaphore* s, uint64_t value, bool signal, hipStream_t stream) {
  if (!s->words || !stream_values_usable(ctx)) return enqueue_semaphore_op(ctx, s, value, signal, stream);
  MipContext::SemaphoreWorkers* w = ctx->semaphore_workers;
  if (!w) {
    w = new (std::nothrow) MipContext::SemaphoreWorkers();
    if (!w) return fail(ctx, MIP_ERR_OUT_OF_MEMORY, "out of host memory");
    w->drm_fd = ctx->drm_fd;
    w->error_word = ctx->h_error + 5;
    w->waiter = std::thread(waiter_main, w);
    w->signaller = std::thread(signaller_main, w);
    ctx->semaphore_workers = w;
  }
  // The stream operation goes first and the request is queued (and the sequence number committed) only when it was accepted:
  // a helper thread never waits for a value no stream will write, and a refused wait does not consume the external signal.
  const unsigned long long seq = (signal ? s->signal_seq : s->wait_seq) + 1;
  if (signal) {
    // Every signal request has its OWN word (a ring indexed by its sequence number): with frames in flight the slots' streams
    // are independent, frame n + 1 may finish before frame n, and one shared word would then show n + 1 to the signaller
    // polling for n — the consumer would be released into a half-written frame n. At most kSignalRing - 1 requests are
    // outstanding per context (the queue is one FIFO), so a ring entry is never rewritten before its request has been served.
    {
      std::unique_lock<std::mutex> lk(w->m);
      if (!w->idle.wait_for(lk, std::chrono::nanoseconds(2 * kSemaphoreWaitNs), [&] { return w->signals.size() + (w->signaller_busy ? 1u : 0u) < kSignalRing - 1; }))
        return fail(ctx, MIP_ERR_TIMEOUT, "%u external-semaphore signals are queued and none has been performed", kSignalRing - 1);
    }
    MIP_HIP(ctx, hipStreamWriteValue64(stream, &s->words[kSignalWord0 + seq % kSignalRing], seq, 0));
    s->signal_seq = seq;
  } else {
    MIP_HIP(ctx, hipStreamWaitValue64(stream, &s->words[0], seq, hipStreamWaitValueGte, ~0ull));
    s->wait_seq = seq;
  }
  {
    std::lock_guard<std::mutex> lk(w->m);
    (signal ? w->signals : w->waits).push_back({s, value, seq});
  }
  (signal ? w->wake_signaller : w->wake_waiter).notify_one();
  return MIP_OK;
}

extern "C" {

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
  unsigned long long* words = nullptr;
  if (drm_handle && stream_values_usable(ctx)) {  // two pinned, device-visible sequence words (stream-value hand-over)
    if (hipHostMalloc((void**)&words, kSemaphoreWordBytes, hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      words = nullptr;  // fall back to host functions for this semaphore
    } else {
      std::memset(words, 0, kSemaphoreWordBytes);
    }
  }
  auto* entry = new (std::nothrow) MipContext::ExternalSemaphore{sem, kind, drm_handle, words, 0ull, 0ull};
  if (!entry) {
    if (sem) (void)hipDestroyExternalSemaphore(sem);
    if (words) (void)hipHostFree(words);
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
  } else if (int32_t rc = enqueue_drm_semaphore(ctx, s, value, false, stream)) {
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
  } else if (int32_t rc = enqueue_drm_semaphore(ctx, s, value, true, stream)) {
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
  if (int32_t rc = interop_drain(ctx)) return rc;
  if (s->sem) MIP_HIP(ctx, hipDestroyExternalSemaphore(s->sem));
  if (s->words) (void)hipHostFree(s->words);
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
