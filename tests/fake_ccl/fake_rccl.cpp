// fake_rccl.cpp — TEST DOUBLE for the five RCCL entry points the library resolves at run time (mip_api.hip, rccl()).
// Lets mip_comm_init / mip_run_sharded run with a world size > 1 on a box with ONE GPU: every rank is a process with
// its own context on the same device; the "collective" moves the chunks through a POSIX shared-memory segment
// (device -> shm slot, barrier, every slot -> device, barrier). Stream-ordered by synchronising the stream, which is
// all a correctness test needs. Selected with MIP_COMM_LIBRARY=<this .so>; never part of the product.
// build: g++ -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.cpp -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

struct Control {
  volatile uint32_t arrived;   // ranks inside the current barrier
  volatile uint32_t generation;
};
struct FakeComm {
  int world, rank;
  size_t slot_bytes;
  unsigned char* base;  // Control, then world slots
  size_t map_bytes;
  char name[64];
};
typedef FakeComm* ncclComm_t;

static const size_t kSlotBytes = 64u << 20;

static double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static bool barrier(FakeComm* c) {
  Control* ctl = reinterpret_cast<Control*>(c->base);
  const uint32_t gen = ctl->generation;
  if (__atomic_add_fetch(&ctl->arrived, 1u, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
    __atomic_store_n(&ctl->arrived, 0u, __ATOMIC_RELEASE);
    __atomic_add_fetch(&ctl->generation, 1u, __ATOMIC_ACQ_REL);
    return true;
  }
  const double t0 = now_s();
  while (__atomic_load_n(&ctl->generation, __ATOMIC_ACQUIRE) == gen) {
    if (now_s() - t0 > 60.0) return false;  // a rank died: fail instead of hanging the test
    usleep(50);
  }
  return true;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/mip_fake_ccl_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
  if (!comm || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  FakeComm* c = new FakeComm();
  c->world = world;
  c->rank = rank;
  c->slot_bytes = kSlotBytes;
  c->map_bytes = 4096 + (size_t)world * kSlotBytes;
  std::strncpy(c->name, id.internal, sizeof c->name - 1);
  int fd = -1;
  const double t0 = now_s();
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); fd = -1; }
  } else {
    while (fd < 0 && now_s() - t0 < 60.0) {  // wait for rank 0 to create and size it
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
  if (!barrier(c)) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
  if (rank == 0) shm_unlink(c->name);  // every rank has it mapped: the name can go
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  munmap(c->base, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  const size_t elem = (type == ncclInt32 || type == ncclUint32) ? 4u : 1u;
  const size_t bytes = count * elem;
  if (bytes > c->slot_bytes) return ncclInvalidArgument;
  unsigned char* slots = c->base + 4096;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // the producer kernel has finished
  if (hipMemcpy(slots + (size_t)c->rank * c->slot_bytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;  // every rank's chunk is in its slot
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy(static_cast<unsigned char*>(recvbuff) + (size_t)r * bytes, slots + (size_t)r * c->slot_bytes, bytes, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;  // nobody overwrites a slot another rank is still reading
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake ccl error"; }

}  // extern "C"
