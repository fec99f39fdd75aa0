// spin_rccl.hip — TEST DOUBLE for the RCCL entry points the library resolves at run time (api_sharded.hip, rccl()),
// like fake_rccl.cpp, but shaped like the real thing where it matters for the kernels that run BESIDE it:
//
//   * ncclAllGather returns at once; the exchange is a KERNEL enqueued on the caller's stream (stream-ordered,
//     asynchronous), so the next frame's shard kernel of another rank / stream meets it on the GPU;
//   * that kernel is a grid of persistent workgroups that SPIN-WAIT ON THE DEVICE for the peers' data (flags in
//     host-mapped shared memory that the peers' kernels write) — the co-tenant shape DESIGN.md section 4 names:
//     spin-waiting workgroups of another queue holding CUs while the instance kernel's tiles wait for each other.
//
// Every rank is a process with its own context on the SAME GPU (the build boxes have one). The segment is POSIX
// shared memory registered with the HIP runtime (hipHostRegisterMapped), so the kernels of every process read and
// write it directly. Waits are bounded (30 s of the realtime counter): a rank that died fails the test instead of
// hanging the box. Selected with MIP_COMM_LIBRARY=<this .so>; never part of the product.
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC spin_rccl.hip -o libspin_rccl.so -lrt
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 4u << 20;  // per rank and parity (the segment is registered, i.e. pinned, as a whole: 64 MiB)
constexpr unsigned long long kSpinTicks = 3000000000ull;  // 30 s at 100 MHz

struct Control {
  // host-side rendezvous of ncclCommInitRank
  volatile uint32_t arrived, generation;
  uint32_t pad0[14];
  // device-side flags, one 64-byte line per rank and parity: the sequence number of the last deposit / last read
  struct alignas(64) Flag { volatile uint32_t seq; uint32_t pad[15]; };
  Flag ready[2][kMaxRanks];
  Flag done[2][kMaxRanks];
  volatile uint32_t failed;  // a bounded device-side wait expired somewhere
};

constexpr size_t kHeaderBytes = (sizeof(Control) + 4095) / 4096 * 4096;

struct SpinComm {
  int world, rank;
  unsigned char* base;    // host mapping: Control, then [2][world] slots
  unsigned char* d_base;  // the same bytes as the device sees them
  size_t map_bytes;
  uint32_t seq;           // collectives issued on this communicator
  uint32_t* d_counters;   // [0] deposits finished, [1] reads finished (workgroups; never reset)
  uint32_t workgroups;
  char name[64];
};

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

bool host_barrier(SpinComm* c) {
  Control* ctl = reinterpret_cast<Control*>(c->base);
  const uint32_t gen = ctl->generation;
  if (__atomic_add_fetch(&ctl->arrived, 1u, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
    __atomic_store_n(&ctl->arrived, 0u, __ATOMIC_RELEASE);
    __atomic_add_fetch(&ctl->generation, 1u, __ATOMIC_ACQ_REL);
    return true;
  }
  const double t0 = now_s();
  while (__atomic_load_n(&ctl->generation, __ATOMIC_ACQUIRE) == gen) {
    if (now_s() - t0 > 60.0) return false;
    usleep(50);
  }
  return true;
}

struct GatherArgs {
  const uint32_t* send;
  uint32_t* recv;
  uint32_t words;        // per rank
  unsigned char* shm;    // device view of the segment
  uint32_t* counters;
  uint32_t seq, world, rank, workgroups;
};

__device__ uint32_t sys_load(const volatile uint32_t* p) {
  return __hip_atomic_load(const_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// true once every rank's flag has reached `want`; false when the bounded wait expired
__device__ bool spin_until_all(Control::Flag* flags, uint32_t world, uint32_t want, Control* ctl) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    bool all = true;
    for (uint32_t r = 0; r < world; ++r) all = all && (int32_t)(sys_load(&flags[r].seq) - want) >= 0;
    if (all) return true;
    if (sys_load(&ctl->failed)) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) {
      __hip_atomic_store(const_cast<uint32_t*>(&ctl->failed), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

__global__ __launch_bounds__(256) void spin_allgather_kernel(const GatherArgs a) {
  Control* ctl = reinterpret_cast<Control*>(a.shm);
  const uint32_t parity = a.seq & 1u;
  uint32_t* slots = reinterpret_cast<uint32_t*>(a.shm + kHeaderBytes);
  const size_t slot_words = kSlotBytes / 4;
  uint32_t* my_slot = slots + ((size_t)parity * kMaxRanks + a.rank) * slot_words;
  __shared__ int s_ok;
  // 0. the slot of this parity is free once every rank has finished READING its previous use (seq - 2)
  if (threadIdx.x == 0) s_ok = (a.seq < 3u) ? 1 : (spin_until_all(ctl->done[parity], a.world, a.seq - 2u, ctl) ? 1 : 0);
  __syncthreads();
  if (!s_ok) return;
  // 1. deposit this rank's chunk
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < a.words; j += a.workgroups * 256u) my_slot[j] = a.send[j];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t before = atomicAdd(&a.counters[0], 1u);
    if (before % a.workgroups == a.workgroups - 1u)  // the last workgroup of this launch to finish its part
      __hip_atomic_store(const_cast<uint32_t*>(&ctl->ready[parity][a.rank].seq), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. EVERY workgroup now spin-waits on the device for the peers — what a collective kernel does
    s_ok = spin_until_all(ctl->ready[parity], a.world, a.seq, ctl) ? 1 : 0;
  }
  __syncthreads();
  if (!s_ok) return;
  __threadfence_system();
  // 3. collect every rank's chunk
  for (uint32_t r = 0; r < a.world; ++r) {
    const uint32_t* src = slots + ((size_t)parity * kMaxRanks + r) * slot_words;
    uint32_t* dst = a.recv + (size_t)r * a.words;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < a.words; j += a.workgroups * 256u) dst[j] = src[j];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t before = atomicAdd(&a.counters[1], 1u);
    if (before % a.workgroups == a.workgroups - 1u)
      __hip_atomic_store(const_cast<uint32_t*>(&ctl->done[parity][a.rank].seq), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

extern "C" {

typedef SpinComm* ncclComm_t;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/mip_spin_ccl_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
  if (!comm || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
  SpinComm* c = new SpinComm();
  c->world = world;
  c->rank = rank;
  c->seq = 0;
  c->map_bytes = kHeaderBytes + (size_t)2 * kMaxRanks * kSlotBytes;
  c->workgroups = 64;  // about what a collective kernel occupies; SPIN_CCL_WORKGROUPS raises it to crowd the chip
  if (const char* env = std::getenv("SPIN_CCL_WORKGROUPS")) c->workgroups = (uint32_t)std::atoi(env) > 0 ? (uint32_t)std::atoi(env) : 64u;
  std::strncpy(c->name, id.internal, sizeof c->name - 1);
  int fd = -1;
  const double t0 = now_s();
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); fd = -1; }
  } else {
    while (fd < 0 && now_s() - t0 < 60.0) {
      fd = shm_open(c->name, O_RDWR, 0600);
      if (fd >= 0) {
        off_t size = lseek(fd, 0, SEEK_END);
        if (size < (off_t)c->map_bytes) { close(fd); fd = -1; }
      }
      if (fd < 0) usleep(1000);
    }
  }
  if (fd < 0) { delete c; return ncclSystemError; }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->base = static_cast<unsigned char*>(p);
  if (hipHostRegister(p, c->map_bytes, hipHostRegisterMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->d_base, p, 0) != hipSuccess ||
      hipMalloc(&c->d_counters, 8) != hipSuccess || hipMemset(c->d_counters, 0, 8) != hipSuccess) {
    munmap(p, c->map_bytes);
    delete c;
    return ncclUnhandledCudaError;
  }
  if (!host_barrier(c)) { (void)hipHostUnregister(p); munmap(p, c->map_bytes); delete c; return ncclSystemError; }
  if (rank == 0) shm_unlink(c->name);
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  (void)hipDeviceSynchronize();
  (void)hipFree(c->d_counters);
  (void)hipHostUnregister(c->base);
  munmap(c->base, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  const size_t bytes = count * ((type == ncclInt32 || type == ncclUint32) ? 4u : 1u);
  if (bytes > kSlotBytes || (bytes & 3u)) return ncclInvalidArgument;
  if (reinterpret_cast<Control*>(c->base)->failed) return ncclSystemError;
  GatherArgs a{};
  a.send = static_cast<const uint32_t*>(sendbuff);
  a.recv = static_cast<uint32_t*>(recvbuff);
  a.words = (uint32_t)(bytes / 4);
  a.shm = c->d_base;
  a.counters = c->d_counters;
  a.seq = ++c->seq;
  a.world = (uint32_t)c->world;
  a.rank = (uint32_t)c->rank;
  a.workgroups = c->workgroups;
  hipLaunchKernelGGL(spin_allgather_kernel, dim3(c->workgroups), dim3(256), 0, stream, a);  // asynchronous, like the real thing
  return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "spin ccl error"; }

}  // extern "C"
