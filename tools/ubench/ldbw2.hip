// L2 -> CU load bandwidth as a function of the ACCESS SHAPE of one wave-instruction (16 B per lane, 1 KiB per instruction):
//   shape R x S: R rows of S contiguous bytes, rows `stride` bytes apart (8 x 128 = a row-major GEMM operand tile at
//   64 k-columns; 1 x 1024 = a blocked / pre-tiled operand).  Modes: global_load_dwordx4 -> VGPR, global_load_lds -> LDS.
// Every workgroup (8 waves) walks its own 48 KB-per-step stream through a buffer that fits the L2s (8 MB), the
// Infinity Cache (64 MB) or neither (1 GB).   Build: hipcc --offload-arch=gfx950 -O3 ldbw2.hip -o ldbw2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void ld_kernel(const char* __restrict__ buf, size_t buf_bytes, int iters, int seg_bytes, int stride,
                                                 float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float4 acc = make_float4(0, 0, 0, 0);
  const int lanes_per_row = seg_bytes / 16;                       // 8 for 128-B segments, 64 for 1 KiB
  const unsigned voff = (unsigned)(lane / lanes_per_row) * (unsigned)stride + (unsigned)(lane % lanes_per_row) * 16u;
  const int rows = 64 / lanes_per_row;                            // rows per instruction
  const size_t span = (size_t)rows * stride;                      // bytes of address space one instruction spans
  const size_t wg_base = ((size_t)blockIdx.x * 2654435761u * 4096u) % (buf_bytes / 2);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      // instruction (wave, i) of step it: consecutive instructions take consecutive row groups; steps advance along the row
      size_t off = wg_base + (size_t)(wave * 6 + i) * span + (size_t)it * seg_bytes;
      off %= (buf_bytes - span - 4096);
      off &= ~(size_t)15;
      const char* g = buf + off;
      if (MODE == 0) {
        const float4 v = *reinterpret_cast<const float4*>(g + voff);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + voff),
                                         (__attribute__((address_space(3))) void*)(smem + ((it % DEPTH) * 48 + wave * 6 + i) * 1024), 16, 0, 0);
      }
    }
    if (MODE == 1) {
      if (DEPTH == 2) __builtin_amdgcn_s_waitcnt(0x0070 | (6 << 0) | (0 << 14) | (15 << 8));    // vmcnt(6): one step in flight
      if (DEPTH == 3) __builtin_amdgcn_s_waitcnt(0x0070 | (12 << 0) | (0 << 14) | (15 << 8));   // vmcnt(12): two steps in flight
      if (DEPTH == 4) __builtin_amdgcn_s_waitcnt(0x0070 | (2 << 0) | (1 << 14) | (15 << 8));    // vmcnt(18): three steps in flight
    }
  }
  if (MODE == 1) {
    __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
    __syncthreads();
    acc.x = *reinterpret_cast<float*>(smem + lane * 4);
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int MODE, int DEPTH>
int run(const char* name, const char* buf, size_t bytes, int seg, int stride, float* sink, int grid) {
  const int iters = 1500;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto k = ld_kernel<MODE, DEPTH>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 48 * 1024 > 160 * 1024 ? 150 * 1024 : 4 * 48 * 1024));
  const int lds = (MODE == 1 ? DEPTH * 48 * 1024 : 1024);
  if (lds > 160 * 1024) return 0;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, buf, bytes, 200, seg, stride, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, buf, bytes, iters, seg, stride, sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double tot = (double)grid * iters * 48 * 1024;
  printf("%-22s buf %5zu MB seg %4d B stride %5d grid %3d: %8.1f GB/s  %6.1f GB/s/CU  %5.1f B/clk/CU\n", name, bytes >> 20, seg, stride,
         grid, tot / ms / 1e6, tot / ms / 1e6 / grid, tot / ms / 1e6 / grid / 2.4);
  return 0;
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 16));
  for (size_t mb : {8, 64, 1024}) {
    char* buf;
    const size_t bytes = mb << 20;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    for (int grid : {256, 32}) {
      const int shapes[][2] = {{128, 1024}, {128, 4096}, {256, 1024}, {256, 4096}, {512, 4096}, {1024, 1024}};
      for (auto& sh : shapes) {
        if (run<0, 2>("vgpr", buf, bytes, sh[0], sh[1], sink, grid)) return 1;
        if (run<1, 2>("lds-dma 1 in flight", buf, bytes, sh[0], sh[1], sink, grid)) return 1;
        if (run<1, 3>("lds-dma 2 in flight", buf, bytes, sh[0], sh[1], sink, grid)) return 1;
      }
    }
    CK(hipFree(buf));
  }
  return 0;
}
