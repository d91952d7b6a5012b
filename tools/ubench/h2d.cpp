// What does one AddSamples upload cost on the host, piece by piece?  (round 6)
// 1.9 MB (30 s of float32 samples) out of PAGEABLE memory, 32 uploads back to back like one caller's batch:
//   a  hipMemcpyAsync(pageable) + stream sync            (what Recognizer::upload does)
//   b  memcpy into a pinned ring, 1 thread               (host time only)
//   c  hipMemcpyAsync(pinned) call                       (host time of the call; completion measured apart)
//   d  hipEventRecord call
//   e  b + c + d per upload, one sync at the end         (the staged form, 1 thread)
//   f  hipHostRegister + hipMemcpyAsync + sync + hipHostUnregister
// Build: hipcc --offload-arch=gfx950 -O2 h2d.cpp -o h2d -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

__global__ void export_kernel(const int* __restrict__ src, volatile int* dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
  __threadfence_system();
}

int main() {
  const size_t n = 480000 * 4, N = 32;
  std::vector<std::vector<float>> src(N, std::vector<float>(n / 4, 0.25f));
  char* dev; CK(hipMalloc((void**)&dev, n * N));
  char* pin; CK(hipHostMalloc((void**)&pin, n * N, hipHostMallocDefault));
  memset(pin, 1, n * N);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev[32]; for (size_t i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = clk::now();
    for (size_t i = 0; i < N; ++i) { CK(hipMemcpyAsync(dev + i * n, src[i].data(), n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
    auto t1 = clk::now();
    printf("a  pageable async + sync      : %7.1f us per upload\n", us(t0, t1) / N);
    t0 = clk::now();
    for (size_t i = 0; i < N; ++i) memcpy(pin + i * n, src[i].data(), n);
    t1 = clk::now();
    printf("b  memcpy -> pinned, 1 thread : %7.1f us per upload (%.1f GB/s)\n", us(t0, t1) / N, n * N / us(t0, t1) / 1e3);
    t0 = clk::now();
    for (size_t i = 0; i < N; ++i) CK(hipMemcpyAsync(dev + i * n, pin + i * n, n, hipMemcpyHostToDevice, s));
    t1 = clk::now();
    CK(hipStreamSynchronize(s));
    auto t2 = clk::now();
    printf("c  async from pinned, call    : %7.1f us per call; all 32 landed %.1f us after the first call (%.1f GB/s)\n", us(t0, t1) / N, us(t0, t2),
           n * N / us(t0, t2) / 1e3);
    t0 = clk::now();
    for (size_t i = 0; i < N; ++i) CK(hipEventRecord(ev[i], s));
    t1 = clk::now();
    CK(hipStreamSynchronize(s));
    printf("d  hipEventRecord call        : %7.1f us\n", us(t0, t1) / N);
    t0 = clk::now();
    for (size_t i = 0; i < N; ++i) {
      memcpy(pin + i * n, src[i].data(), n);
      CK(hipMemcpyAsync(dev + i * n, pin + i * n, n, hipMemcpyHostToDevice, s));
      CK(hipEventRecord(ev[i], s));
    }
    t1 = clk::now();
    CK(hipStreamSynchronize(s));
    t2 = clk::now();
    printf("e  staged, 1 thread           : %7.1f us per upload on the host, +%.1f us until the last one landed\n", us(t0, t1) / N, us(t1, t2));
    t0 = clk::now();
    for (size_t i = 0; i < N; ++i) {
      CK(hipHostRegister(src[i].data(), n, hipHostRegisterDefault));
      CK(hipMemcpyAsync(dev + i * n, src[i].data(), n, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      CK(hipHostUnregister(src[i].data()));
    }
    t1 = clk::now();
    printf("f  register + copy + unregister: %6.1f us per upload\n", us(t0, t1) / N);
    {
      hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      hipEvent_t tmp[32];
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&tmp[i], hipEventDisableTiming));
      t1 = clk::now();
      printf("h  hipEventCreateWithFlags    : %7.1f us\n", us(t0, t1) / N);
      for (size_t i = 0; i < N; ++i) CK(hipEventRecord(tmp[i], s));
      CK(hipStreamSynchronize(s));
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) CK(hipStreamWaitEvent(s2, tmp[i], 0));
      t1 = clk::now();
      printf("i  hipStreamWaitEvent (done)  : %7.1f us\n", us(t0, t1) / N);
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) CK(hipEventSynchronize(tmp[i]));
      t1 = clk::now();
      printf("j  hipEventSynchronize (done) : %7.1f us\n", us(t0, t1) / N);
      CK(hipStreamSynchronize(s2));
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) CK(hipEventDestroy(tmp[i]));
      t1 = clk::now();
      printf("k  hipEventDestroy            : %7.1f us\n", us(t0, t1) / N);
      // the staged form with everything the recognizer adds per upload: 2 pieces, 3 records, a wait on another stream
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) {
        for (int h = 0; h < 2; ++h) {
          memcpy(pin + i * n + h * (n / 2), (const char*)src[i].data() + h * (n / 2), n / 2);
          CK(hipMemcpyAsync(dev + i * n + h * (n / 2), pin + i * n + h * (n / 2), n / 2, hipMemcpyHostToDevice, s));
          CK(hipEventRecord(ev[i], s));
        }
        CK(hipEventRecord(ev[i], s));
      }
      t1 = clk::now();
      for (size_t i = 0; i < N; ++i) CK(hipStreamWaitEvent(s2, ev[i], 0));
      t2 = clk::now();
      CK(hipStreamSynchronize(s2));
      auto t3 = clk::now();
      printf("l  staged, 2 pieces + records : %7.1f us per upload; 32 waits on the consumer stream %.1f us; consumer free %.1f us later\n",
             us(t0, t1) / N, us(t1, t2), us(t2, t3));
      CK(hipStreamDestroy(s2));
    }
    {
      // device -> host, small: what the engine's read-backs cost (decoder length: 3 x ~132 B; the ids: 43 KB)
      std::vector<char> pageable(65536);
      for (size_t bytes : {(size_t)4, (size_t)132, (size_t)43008}) {
        t0 = clk::now();
        for (size_t i = 0; i < N; ++i) { CK(hipMemcpyAsync(pageable.data(), dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        t1 = clk::now();
        for (size_t i = 0; i < N; ++i) { CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        t2 = clk::now();
        printf("m  D2H %6zu B + sync         : pageable %6.1f us   pinned %6.1f us\n", bytes, us(t0, t1) / N, us(t1, t2) / N);
      }
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) {
        hipLaunchKernelGGL(export_kernel, dim3(1), dim3(256), 0, s, (const int*)dev, (volatile int*)pin, 33);
        CK(hipEventRecord(ev[0], s));
        CK(hipEventSynchronize(ev[0]));
      }
      t1 = clk::now();
      printf("n  kernel -> mapped pinned + event sync (132 B) : %6.1f us\n", us(t0, t1) / N);
      t0 = clk::now();
      for (size_t i = 0; i < N; ++i) {
        hipLaunchKernelGGL(export_kernel, dim3(42), dim3(256), 0, s, (const int*)dev, (volatile int*)pin, 10752);
        CK(hipStreamSynchronize(s));
      }
      t1 = clk::now();
      printf("o  kernel -> mapped pinned + stream sync (43 KB): %6.1f us\n", us(t0, t1) / N);
    }
    t0 = clk::now();
    CK(hipMemcpy(dev, pin, n * N, hipMemcpyHostToDevice));
    t1 = clk::now();
    printf("g  one 61 MB copy from pinned : %7.1f us (%.1f GB/s)\n\n", us(t0, t1), n * N / us(t0, t1) / 1e3);
  }
  return 0;
}
