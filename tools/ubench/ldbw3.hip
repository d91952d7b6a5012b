// Do the two L2 -> CU operand paths ADD?  LDS-DMA (global_load_lds, 1 KiB per wave-instruction) tops out at ~79 GB/s per CU
// and plain global_load_dwordx4 -> VGPR at 58-72 GB/s with 6 loads per wave in flight (ldbw2.hip).  This bench keeps a
// software-pipelined ring of DEPTH 16-byte loads in flight per wave on the VGPR path (what a GEMM that takes its private
// operand straight into registers would do) and runs it alone, beside LDS-DMA waves, and mixed inside every wave.
//   mode 0: all 8 waves VGPR ring        mode 1: all 8 waves LDS-DMA (6 pieces per step, one step in flight)
//   mode 2: waves 0-3 LDS-DMA (12 pieces per step), waves 4-7 VGPR ring
//   mode 3: every wave: 6 DMA pieces per step + a VGPR ring
// All accesses are 1 KiB contiguous per wave-instruction from an L2-resident buffer shared by all workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 ldbw3.hip -o ldbw3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// register-destination load the compiler does not track: the ring below waits with its own counted s_waitcnt
__device__ __forceinline__ void gld(f4& r, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_use(f4& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory"); }

__device__ __forceinline__ void dma(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void ld_kernel(const char* __restrict__ buf, size_t buf_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t mask = buf_bytes - 1;                                  // buffer sizes are powers of two
  const size_t wg_base = ((size_t)blockIdx.x * 2654435761u * 4096u) & mask;
  const bool dma_wave = (MODE == 1) || (MODE == 3) || (MODE == 2 && wave < 4);
  const bool reg_wave = (MODE == 0) || (MODE == 3) || (MODE == 2 && wave >= 4);
  const int per = (MODE == 2) ? 12 : 6;                               // DMA pieces per wave and step
  f4 ring[DEPTH];
  float4 acc = make_float4(0, 0, 0, 0);
  size_t rpos = (wg_base + (buf_bytes >> 1) + (size_t)wave * 1024) & mask & ~(size_t)1023;
  if (reg_wave) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      gld(ring[i], buf + rpos + lane * 16);
      rpos = (rpos + 8 * 1024) & mask;
    }
  }
  for (int it = 0; it < iters; ++it) {
    if (dma_wave) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        if (i < per) {
          size_t off = (wg_base + ((size_t)(it * 8 + wave) * 12 + i) * 1024) & mask & ~(size_t)1023;
          dma(buf + off + lane * 16, smem + ((it & 1) * 48 + (MODE == 2 ? wave * 12 : wave * 6) + i) * 1024);
        }
      }
    }
    if (reg_wave) {
      // 12 (mode 0, 2) or 6 (mode 3) ring positions per step: consume the oldest load, issue a new one in its place
#pragma unroll
      for (int j = 0; j < (MODE == 3 ? 6 : 12); ++j) {
        const int slot = j % DEPTH;
        // the oldest of the DEPTH loads in flight (in mode 3 the 6 DMA pieces of this step were issued after it as well)
        wait_use<(MODE == 3 ? DEPTH - 1 + 6 : DEPTH - 1)>(ring[slot]);
        acc.x += ring[slot].x; acc.y += ring[slot].y;
        acc.z += ring[slot].z; acc.w += ring[slot].w;
        gld(ring[slot], buf + rpos + lane * 16);
        rpos = (rpos + 8 * 1024) & mask;
      }
    }
    if (MODE == 3) __builtin_amdgcn_s_waitcnt(0x0070 | ((DEPTH + 6) & 15) | (((DEPTH + 6) >> 4) << 14) | (15 << 8));   // last step's DMA pieces
    if (dma_wave && !reg_wave) __builtin_amdgcn_s_waitcnt(0x0070 | (per & 15) | (0 << 14) | (15 << 8));   // one step in flight
  }
  __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
  __syncthreads();
  float s = *reinterpret_cast<float*>(smem + lane * 4) + acc.x + acc.y + acc.z + acc.w;
  if (reg_wave)
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) s += ring[i].x;
  if (s == 123.456f) sink[0] = s;
}

template <int MODE, int DEPTH>
int run(const char* name, const char* buf, size_t bytes, float* sink, int grid) {
  const int iters = 1500;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("[%s depth %d]\n", name, DEPTH); fflush(stdout);
  auto k = ld_kernel<MODE, DEPTH>;
  const int lds = 96 * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, buf, bytes, 200, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, buf, bytes, iters, sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  // KiB per workgroup and step
  const double dma_k = MODE == 1 ? 48 : (MODE == 2 ? 48 : (MODE == 3 ? 48 : 0));
  const double reg_k = MODE == 0 ? 96 : (MODE == 2 ? 48 : (MODE == 3 ? 48 : 0));
  const double tot = (double)grid * iters * (dma_k + reg_k) * 1024;
  printf("%-44s depth %2d buf %3zu MB grid %3d: %8.1f GB/s  %6.1f GB/s/CU (dma %4.1f + vgpr %4.1f)\n", name, DEPTH, bytes >> 20, grid,
         tot / ms / 1e6, tot / ms / 1e6 / grid, dma_k / (dma_k + reg_k) * tot / ms / 1e6 / grid, reg_k / (dma_k + reg_k) * tot / ms / 1e6 / grid);
  fflush(stdout);
  return 0;
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 16));
  for (size_t mb : {4, 64}) {
    char* buf;
    const size_t bytes = mb << 20;
    CK(hipMalloc(&buf, bytes + 8192));
    CK(hipMemset(buf, 1, bytes + 8192));
    for (int grid : {256, 32}) {
      if (run<1, 4>("lds-dma only", buf, bytes, sink, grid)) return 1;
      if (run<0, 6>("vgpr ring only", buf, bytes, sink, grid)) return 1;
      if (run<0, 12>("vgpr ring only", buf, bytes, sink, grid)) return 1;
      if (run<2, 6>("waves 0-3 dma x12 | waves 4-7 vgpr ring", buf, bytes, sink, grid)) return 1;
      if (run<2, 12>("waves 0-3 dma x12 | waves 4-7 vgpr ring", buf, bytes, sink, grid)) return 1;
      if (run<3, 6>("every wave: dma x6 + vgpr ring x6", buf, bytes, sink, grid)) return 1;
    }
    CK(hipFree(buf));
  }
  return 0;
}
