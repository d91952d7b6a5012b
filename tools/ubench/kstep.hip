// What bounds a GEMM k-step on one CU: LDS-DMA (HBM/L2 -> LDS), fragment reads (LDS -> VGPR) or the MFMAs — and which of
// them overlap.  One workgroup per CU runs `iters` k-steps of a tile shape with any subset of the three activities:
//   D  6 or 8 (or 16) global_load_lds_dwordx4 per wave per step into a 3-stage ring (counted vmcnt, two stages in flight)
//   R  ds_read_b128 fragment reads of the stage that landed
//   M  v_mfma_f32_32x32x16_f16 on the fragments (constant registers when R is off)
// Shapes:   A: 8 waves, 256x128 tile, k-step 64, wave 64x64    (48 KB DMA, 128 KB reads, 16 MFMA / wave / step)
//           B: 8 waves, 256x256 tile, k-step 32, wave 128x64   (32 KB DMA,  96 KB reads, 16 MFMA / wave / step)
//           C: 4 waves, 256x256 tile, k-step 32, wave 128x128  (32 KB DMA,  64 KB reads, 32 MFMA / wave / step)
// 'D' = all DMA pieces of a step issued at its start, 'd' = one piece behind each of the first MFMAs.
// One s_barrier per step.  Operands come from an 8 MB (L2-resident) buffer.
// Build: hipcc --offload-arch=gfx950 -O3 kstep.hip -o kstep
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NW, int PIECES, int RA, int RB, int KC, bool D, bool R, bool M, bool IL>
__global__ __launch_bounds__(NW * 64) void k(const char* __restrict__ buf, size_t bytes, int iters, float* sink) {
  // RA / RB: 32-row fragment blocks of A / B per wave; reads per step = (RA + RB) * 4 (k64 = 4 chunks of 16), MFMAs = RA * RB * 4
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int STAGE = NW * PIECES * 1024;
  f16v acc[RA][RB];
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  h8 fa[RA], fb[RB];
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[a][e] = (_Float16)(0.001f * (lane + a + e));
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) fb[b][e] = (_Float16)(0.002f * (lane + b + e));
  const size_t base = ((size_t)blockIdx.x * 40503u * 4096u) % (bytes / 2);
  for (int it = 0; it < iters; ++it) {
    auto piece = [&](int i) __attribute__((always_inline)) {
      size_t off = (base + (size_t)(wave * PIECES + i) * 1024 + (size_t)it * STAGE) % (bytes - STAGE - 4096);
      off &= ~(size_t)1023;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + (it % 3) * STAGE + (wave * PIECES + i) * 1024), 16, 0, 0);
    };
    if (D && !IL) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) piece(i);
    }
    int q = 0;
    const char* st = smem + ((it + 1) % 3) * STAGE;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if (R) {
#pragma unroll
        for (int a = 0; a < RA; ++a) fa[a] = *reinterpret_cast<const h8*>(st + ((wave * 7 + a * 4 + kc) * 1024 + lane * 16) % STAGE);
#pragma unroll
        for (int b = 0; b < RB; ++b) fb[b] = *reinterpret_cast<const h8*>(st + ((wave * 5 + 29 + b * 4 + kc) * 1024 + lane * 16) % STAGE);
      }
      if (M) {
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
          for (int b = 0; b < RB; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
            if (D && IL && q < PIECES) {                       // one DMA piece behind each of the first MFMAs, as the GEMM kernels issue them
              __builtin_amdgcn_sched_barrier(0);
              piece(q++);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      } else if (R) {
#pragma unroll
        for (int a = 0; a < RA; ++a) acc[a][0][0] += (float)fa[a][0];
#pragma unroll
        for (int b = 0; b < RB; ++b) acc[0][b][1] += (float)fb[b][0];
      }
    }
    if (D) {
      if (PIECES == 6) __builtin_amdgcn_s_waitcnt(0x0070 | 12 | (15 << 8));
      if (PIECES == 4) __builtin_amdgcn_s_waitcnt(0x0070 | 8 | (15 << 8));
      if (PIECES == 8) __builtin_amdgcn_s_waitcnt(0x0070 | 0 | (1 << 14) | (15 << 8));      // vmcnt(16)
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  if (s == 123.456f) sink[0] = s;
}

template <int NW, int PIECES, int RA, int RB, int KC, bool D, bool R, bool M, bool IL>
int run(const char* shape, const char* buf, size_t bytes, float* sink, int grid) {
  const int iters = 4000;
  auto kk = k<NW, PIECES, RA, RB, KC, D, R, M, IL>;
  const int lds = 3 * NW * PIECES * 1024;
  CK(hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(kk, dim3(grid), dim3(NW * 64), lds, 0, buf, bytes, 100, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(kk, dim3(grid), dim3(NW * 64), lds, 0, buf, bytes, iters, sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1000.0 / iters;
  const double mflop = 2.0 * NW * RA * RB * 32 * 32 * 16 * KC / 1e6;
  printf("%s grid %3d  %c%c%c  %.3f us/step  %.3f us per 4.19 MFLOP", shape, grid, D ? (IL ? 'd' : 'D') : '-', R ? 'R' : '-', M ? 'M' : '-', us, us * 4.194304 / mflop);
  if (M) printf("  (%.0f TFLOP/s on 256 CUs)", mflop / us * 256 / 1e6 * 1e6);
  printf("\n");
  return 0;
}

#define ALL(NW, P, RA, RB, KC, name)                                              \
  if (run<NW, P, RA, RB, KC, true, false, false, false>(name, buf, bytes, sink, grid)) return 1; \
  if (run<NW, P, RA, RB, KC, false, true, false, false>(name, buf, bytes, sink, grid)) return 1; \
  if (run<NW, P, RA, RB, KC, false, false, true, false>(name, buf, bytes, sink, grid)) return 1; \
  if (run<NW, P, RA, RB, KC, true, true, false, false>(name, buf, bytes, sink, grid)) return 1;  \
  if (run<NW, P, RA, RB, KC, true, false, true, false>(name, buf, bytes, sink, grid)) return 1;  \
  if (run<NW, P, RA, RB, KC, true, false, true, true>(name, buf, bytes, sink, grid)) return 1;   \
  if (run<NW, P, RA, RB, KC, false, true, true, false>(name, buf, bytes, sink, grid)) return 1;  \
  if (run<NW, P, RA, RB, KC, true, true, true, false>(name, buf, bytes, sink, grid)) return 1;   \
  if (run<NW, P, RA, RB, KC, true, true, true, true>(name, buf, bytes, sink, grid)) return 1;

int main() {
  float* sink;
  CK(hipMalloc(&sink, 16));
  char* buf;
  const size_t bytes = (size_t)8 << 20;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  for (int grid : {256, 32}) {
    ALL(8, 6, 2, 2, 4, "A 8w 256x128 k64 w64x64  ")
    ALL(8, 4, 4, 2, 2, "B 8w 256x256 k32 w128x64 ")
    ALL(4, 8, 4, 4, 2, "C 4w 256x256 k32 w128x128")
  }
  return 0;
}
