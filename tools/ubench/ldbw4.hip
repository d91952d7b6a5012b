// What does the ACCESS PATTERN of the fused encoder launch's weight stream cost?  (round 6)
// ffn_fused_kernel streams pre-tiled weight images straight into registers: every workgroup reads the SAME image in the SAME
// order at (nearly) the same time, wave w of a workgroup walks its own slice — slices 32 KiB (W1 / W2) or 64 KiB (Wo / Wq)
// apart — 1 KiB per wave-instruction, PF fragments in flight.  The phase timeline (tools/ffn_timing.sh) shows 30-48 B/clk per CU
// where ldbw3 (every workgroup at its own place, waves adjacent) reaches 114-126 GB/s = 55-60 B/clk.  This bench reproduces the
// kernel's pattern and varies one thing at a time:
//   wave stride  : 32 KiB | 64 KiB | +4 KiB skew | +1 KiB skew   (do power-of-two strides pile onto few L2 channels?)
//   rotation     : every workgroup the same order | workgroups of an XCD start at R different places of the image
//   depth        : 8 | 16 fragments in flight per wave
// Build: hipcc --offload-arch=gfx950 -O3 ldbw4.hip -o ldbw4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void gld(f4& r, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_use(f4& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory"); }

// image: NSEG segments; a segment = 8 wave slices of `frags` KiB each, slice s of wave w at w * wstride; segments seg_stride apart.
// A workgroup walks segments (seg + rot) % NSEG, fragment by fragment; `reps` passes over the image.
template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ img, int nseg, int frags, size_t wstride, size_t seg_stride,
                                                     int rot_groups, int reps, float* sink, size_t fstride = 1024, size_t wg_private = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rot = rot_groups > 1 ? (int)((blockIdx.x >> 3) % (unsigned)rot_groups) * (nseg / rot_groups) : 0;
  // the next fragment to request: (segment counter, fragment inside the slice) advanced incrementally (no divisions in the loop)
  int nseg_i = 0, nf = 0;
  auto next = [&]() -> const char* {
    int seg = nseg_i + rot;
    seg = seg >= nseg ? seg - nseg : seg;
    const char* p = img + (size_t)blockIdx.x * wg_private + (size_t)seg * seg_stride + (size_t)wave * wstride + (size_t)nf * fstride + lane * 16;
    if (++nf == frags) { nf = 0; if (++nseg_i == nseg) nseg_i = 0; }
    return p;
  };
  f4 ring[DEPTH];
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) gld(ring[i], next());
  const int n = nseg * frags * reps;
  for (int g = 0; g < n; g += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      wait_use<DEPTH - 1>(ring[j]);
      acc.x += ring[j].x; acc.y += ring[j].y; acc.z += ring[j].z; acc.w += ring[j].w;
      gld(ring[j], next());
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = acc.x + acc.y + acc.z + acc.w;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) s += ring[i].x;
  if (s == 123.456f) sink[0] = s;
}

template <int DEPTH>
int run(const char* name, const char* img, int nseg, int frags, size_t wstride, size_t seg_stride, int rot_groups, float* sink, int grid,
        size_t fstride = 1024, size_t wg_private = 0) {
  const int reps = 24;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(512), 0, 0, img, nseg, frags, wstride, seg_stride, rot_groups, 2, sink, fstride, wg_private);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(512), 0, 0, img, nseg, frags, wstride, seg_stride, rot_groups, reps, sink, fstride, wg_private);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)grid * 8 * nseg * frags * 1024.0 * reps;
  printf("%-58s depth %2d rot %2d grid %3d: %7.1f GB/s/CU  (%6.0f GB/s)\n", name, DEPTH, rot_groups, grid, bytes / ms / 1e6 / grid, bytes / ms / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 16));
  char* img;
  const size_t bytes = (size_t)64 << 20;
  CK(hipMalloc(&img, bytes));
  CK(hipMemset(img, 1, bytes));
  for (int grid : {250, 32}) {
    // the kernel's pattern: 16 segments of 8 wave slices x 32 KiB (4 MiB per pass), every workgroup the same image in the same order
    if (run<8>("kernel pattern: wave stride 32 KiB, shared image", img, 16, 32, 32 << 10, 256 << 10, 1, sink, grid)) return 1;
    if (run<4>("kernel pattern: wave stride 32 KiB, shared image", img, 16, 32, 32 << 10, 256 << 10, 1, sink, grid)) return 1;
    // waves ADJACENT: wave w reads fragment 8 f + w of a contiguous 256 KiB segment (a workgroup step = 8 KiB contiguous)
    if (run<8>("adjacent waves (8 KiB contiguous per step), shared image", img, 16, 32, 1 << 10, 256 << 10, 1, sink, grid, 8 << 10)) return 1;
    // every workgroup its OWN 128 KiB image (32 MB in all: L2-resident per XCD), kernel pattern inside it
    if (run<8>("private 128 KiB image per workgroup, wave stride 16 KiB", img, 1, 16, 16 << 10, 0, 1, sink, grid, 1024, 128 << 10)) return 1;
    if (run<8>("private 128 KiB image per workgroup, adjacent waves", img, 1, 16, 1 << 10, 0, 1, sink, grid, 8 << 10, 128 << 10)) return 1;
    // a SMALL shared image (512 KiB: one chunk) — does the footprint matter?
    if (run<8>("kernel pattern, shared 512 KiB image", img, 2, 32, 32 << 10, 256 << 10, 1, sink, grid)) return 1;
  }
  return 0;
}
