// stream_value_probe.cpp — are hipStreamWaitValue64 / hipStreamWriteValue64 usable on this runtime for the semaphore hand-over
// (row f-2), on which memory, and what do they cost in a stream of small kernels? (round 4; output kept in
// profiles/r04_external_semaphore_handover.txt)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void tiny(unsigned long long* p) { if (p && threadIdx.x == 12345) *p = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  int can = -1;
  hipError_t e = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  std::printf("hipDeviceAttributeCanUseStreamWaitValue: %s, value %d\n", hipGetErrorString(e), can);
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int kind = 0; kind < 2; ++kind) {
    unsigned long long* w = nullptr;
    if (kind == 0) e = hipHostMalloc((void**)&w, 64, hipHostMallocMapped);
    else e = hipExtMallocWithFlags((void**)&w, 8, hipMallocSignalMemory);
    std::printf("%s: alloc %s\n", kind == 0 ? "hipHostMalloc(mapped)" : "hipExtMallocWithFlags(SignalMemory)", hipGetErrorString(e));
    if (e != hipSuccess) continue;
    unsigned long long* hw = w;  // both kinds are host-accessible pointers
    *hw = 0;
    // functional: the stream must not pass the wait before the host writes
    std::atomic<int> done{0};
    e = hipStreamWaitValue64(st, w, 5, hipStreamWaitValueGte, ~0ull);
    std::printf("  hipStreamWaitValue64: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { (void)hipGetLastError(); continue; }
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, nullptr);
    e = hipStreamWriteValue64(st, w + 1 - kind, 77, 0);  // host memory: a second word; signal memory has one word: overwrite it
    std::printf("  hipStreamWriteValue64: %s\n", hipGetErrorString(e));
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    const bool passed_early = hipStreamQuery(st) == hipSuccess;
    const double t0 = now_us();
    *hw = 5;
    (void)hipStreamSynchronize(st);
    std::printf("  blocked until the host wrote: %s; released %.1f us after the host's store; write landed: %llu\n", passed_early ? "NO" : "yes",
                now_us() - t0, kind == 0 ? hw[1] : hw[0]);
    // cost per frame: wait (already satisfied) + kernel + write, against the bare kernel
    const int frames = 5000;
    *hw = ~0ull >> 1;
    (void)hipStreamSynchronize(st);
    double t = now_us();
    for (int k = 0; k < frames; ++k) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, nullptr);
    (void)hipStreamSynchronize(st);
    const double bare = (now_us() - t) / frames;
    unsigned long long* target = kind == 0 ? w + 1 : w;
    t = now_us();
    for (int k = 0; k < frames; ++k) {
      if (kind == 0) (void)hipStreamWaitValue64(st, w, (unsigned long long)k, hipStreamWaitValueGte, ~0ull);
      hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, nullptr);
      (void)hipStreamWriteValue64(st, target, (unsigned long long)k + 100, 0);
    }
    (void)hipStreamSynchronize(st);
    std::printf("  per frame: bare kernel %.2f us, %skernel + write %.2f us\n", bare, kind == 0 ? "wait + " : "", (now_us() - t) / frames);
  }
  return 0;
}
