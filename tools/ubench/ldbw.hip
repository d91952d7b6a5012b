// L2 -> CU load-bandwidth microbenchmark: every workgroup streams tiles of an L2/MALL-resident
// buffer with (0) global_load_dwordx4 -> VGPR, (1) global_load_lds_dwordx4 -> LDS.
// Build: hipcc --offload-arch=gfx950 -O3 ldbw.hip -o ldbw ; run: ./ldbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int NW, int ROWS_STRIDE>
__global__ __launch_bounds__(NW * 64) void ld_kernel(const char* __restrict__ buf, size_t buf_bytes, int iters,
                                                     int row_bytes, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float4 acc = make_float4(0, 0, 0, 0);
  // each wave-instruction: 8 rows x 128 B (rows row_bytes apart), like a GEMM operand tile
  const unsigned voff = (lane >> 3) * row_bytes + (lane & 7) * 16;
  size_t base = ((size_t)blockIdx.x * 7919u * 4096u) % (buf_bytes - 64u * 1024u * 1024u / 16);
  for (int it = 0; it < iters; ++it) {
    // 6 loads per wave per step (= 48 KB per 8-wave block)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const char* g = buf + ((base + (size_t)(wave * 6 + i) * 8 * row_bytes + (size_t)it * 128) % (buf_bytes - (1 << 20)));
      if (MODE == 0) {
        const float4 v = *reinterpret_cast<const float4*>(g + voff);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + voff),
                                         (__attribute__((address_space(3))) void*)(smem + ((it & 1) * 48 + wave * 6 + i) * 1024), 16, 0, 0);
      }
    }
    if (MODE == 1 && (it & 3) == 3) __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));   // vmcnt(0)
  }
  if (MODE == 1) {
    __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
    __syncthreads();
    acc.x = *reinterpret_cast<float*>(smem + lane * 4);
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int MODE>
int run(const char* name, const char* buf, size_t bytes, int row_bytes, float* sink, int grid) {
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto k = ld_kernel<MODE, 8, 0>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 96 * 1024, 0, buf, bytes, 200, row_bytes, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 96 * 1024, 0, buf, bytes, iters, row_bytes, sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double tot = (double)grid * iters * 48 * 1024;
  printf("%-34s buf %5zu MB row %5d B grid %4d: %8.1f GB/s  (%.1f GB/s/CU, %.1f B/clk/CU @2.4GHz)\n", name, bytes >> 20,
         row_bytes, grid, tot / ms / 1e6, tot / ms / 1e6 / 256, tot / ms / 1e6 / 256 / 2.4);
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
    for (int rb : {1024, 4096}) {
      if (run<0>("global_load_dwordx4 -> VGPR", buf, bytes, rb, sink, 256)) return 1;
      if (run<1>("global_load_lds_dwordx4 -> LDS", buf, bytes, rb, sink, 256)) return 1;
    }
    CK(hipFree(buf));
  }
  return 0;
}
