// k_gemm_big.hip — 256 x {192, 256} tile GEMM for the wide projections with f16 results (QKV: N = 1536, FFN-up:
// N = 2048, the 16 decoder K/V projections: N = 16384), K = 512..576:
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias  [* q-scale on the first columns] [ReLU]  -> f16 (row-major or blocked layout)
//
// (the MatMul + Add [+ Mul] [+ Relu] nodes of the graph InferenceSession.Run executes,
// AliParaformerAsr/OfflineProjOfParaformer.cs:68).
//
// Why a second kernel next to gemm_f16_pp3 (k_gemm.hip): with K = 512 a 256 x 128 tile has only 8 k-steps, stages
// 48 KB per 4.2 MFLOP and pays its prologue / epilogue / tile-boundary costs every 8 steps (measured 1.46 us per
// k-step against 1.05 us for the K = 2048 FFN-down).  A 256 x 256 (256 x 192) tile stages 64 KB (56 KB) per 8.4
// (6.3) MFLOP — 1.5 x less LDS feed per flop — and halves the number of tile boundaries.  Structure (shared with
// k_gemm_rc.hip, which showed that at this feed-bound operating point a plain 2-stage ring matches the ping-pong
// pipeline per k-step): ONE tile per 512-thread workgroup (8 waves as 4 x 2, wave tile 64 x 128 / 64 x 96 =
// 2 x NJ MFMA 32x32x16 blocks, 128 / 96 accumulator registers), operands HBM -> LDS with global_load_lds_dwordx4
// into a 2-stage ring consumed behind a counted vmcnt and raw s_barriers, B fragments of a k-step loaded up front,
// A fragments streamed one k-sub ahead, the next stage's DMA pieces slotted between the MFMAs once the stage's
// reads have retired.  Epilogue: bias / scale / ReLU in registers, f16 tile through LDS (the ring is free: one
// tile per workgroup), whole 384 / 512-byte row segments (or whole 512-byte blocks of the blocked layout) to HBM.
// Workgroups are dealt to XCDs in contiguous runs with n fastest, so an A panel is shared in one private L2.
#include "kernels.h"

#include <cstdlib>
#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef h4 __attribute__((may_alias)) h4a;
typedef h8 __attribute__((may_alias)) h8a;

struct BigDev {
  const half_t* A; const half_t* W; const float* bias; half_t* out;
  int lda, ldw, ldc;
  int M, N, K, tiles_m, tiles_n;
  int relu, scale_cols, blocked;
  float scale;
  int abl;   // timing experiments only (PF_BIG_ABL): 1 no steady-state DMA, 2 no MFMA, 4 no result stores, 8 no fragment reads
};

constexpr int BG_BM = 256, BG_BK = 64, BG_ROWB = BG_BK * 2;
constexpr int bg_stage(int nj) { return (BG_BM + 64 * nj) * BG_ROWB; }          // A tile + W tile of one k-step
constexpr int bg_xrow(int nj) { return 64 * nj * 2 + 16; }                       // f16 epilogue row + 16-byte skew
constexpr int bg_lds(int nj) { return 2 * bg_stage(nj) > BG_BM * bg_xrow(nj) ? 2 * bg_stage(nj) : BG_BM * bg_xrow(nj); }

__device__ __forceinline__ void bg_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void bg_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// result stores are write-through and dropped from the XCD's L2 (sc1): 66 MB of results per launch would otherwise
// evict the W panel and the A panels the other tiles of this XCD are re-reading (k_gemm.hip measured the same)
__device__ __forceinline__ void bg_store16(void* p, h8 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bg_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// NJ: 32-column MFMA blocks per wave (4 -> 256-column tiles, 3 -> 192-column tiles)
template <int NJ>
__global__ __launch_bounds__(512, 1) void gemm_big_kernel(BigDev p) {
  constexpr int BN = 64 * NJ, A_BYTES = BG_BM * BG_ROWB, STAGE = bg_stage(NJ), XROW = bg_xrow(NJ);
  constexpr int A_PW = 4, W_PW = NJ, LPS = A_PW + W_PW;      // 1 KiB DMA pieces per wave per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lh = lane >> 5;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 1) & 7; };

  // ---- tile of this workgroup: XCD b % 8 gets a contiguous run of tiles, n fastest
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BG_BM, n0 = tn * BN;
  const int nk = p.K / BG_BK;

  // ---- LDS-DMA source offsets (bytes, per lane)
  const int srow = lane >> 3, schunk = lane & 7;
  unsigned a_vo[A_PW], w_vo[W_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = (wave + 8 * i) * 8 + srow;
    a_vo[i] = (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < W_PW; ++i) {
    const int row = (wave + 8 * i) * 8 + srow;
    w_vo[i] = (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  const char* a_base = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
  const char* w_base = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw);
  auto issue_piece = [&](int k, int buf, int piece) __attribute__((always_inline)) {
    char* st = smem + buf * STAGE + wave * 1024;
    if (piece < A_PW) bg_glds16(a_base + (size_t)k * (BG_BK * 2) + a_vo[piece < A_PW ? piece : 0], st + piece * 8192);
    else bg_glds16(w_base + (size_t)k * (BG_BK * 2) + w_vo[piece >= A_PW ? piece - A_PW : 0], st + A_BYTES + (piece - A_PW) * 8192);
  };

  // ---- fragment read offsets inside a stage (bytes)
  unsigned fa[4][2], fb[4][NJ];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + (lane & 31);
      fa[s][i] = (unsigned)(ra * BG_ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int rb = wn * (32 * NJ) + j * 32 + (lane & 31);
      fb[s][j] = (unsigned)(A_BYTES + rb * BG_ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  }

  f16x acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- main loop: 2-stage ring, stage k+1 in flight across the barriers of step k
#pragma unroll
  for (int q = 0; q < LPS; ++q) issue_piece(0, 0, q);
  if (nk > 1) {
#pragma unroll
    for (int q = 0; q < LPS; ++q) issue_piece(1, 1, q);
  }
  for (int k = 0; k < nk; ++k) {
    if (k + 1 < nk) bg_wait_vmcnt<LPS>(); else bg_wait_vmcnt<0>();     // this wave's pieces of stage k have landed
    __builtin_amdgcn_s_barrier();                                       // ... and everybody else's
    const char* rd = smem + (k & 1) * STAGE;
    const bool more = k + 2 < nk;
    // fragments are streamed one k-sub ahead of the MFMAs that use them (two register sets), so the LDS reads of
    // sub s+1 run under the 2*NJ MFMAs of sub s; the stage is released (barrier) once the reads of the LAST sub have
    // retired, and the next-but-one stage's DMA pieces go out between the MFMAs of that last sub
    h8 a0[2] = {}, b0[NJ] = {}, a1[2] = {}, b1[NJ] = {};
    auto load = [&](int s, h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
      if (p.abl & 8) return;
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = *(const h8*)(rd + fb[s][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const h8*)(rd + fa[s][i]);
    };
    auto mma = [&](h8 (&af)[2], h8 (&bf)[NJ], bool dma) __attribute__((always_inline)) {
      int piece = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (!(p.abl & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          if (dma && piece < LPS) {
            __builtin_amdgcn_sched_barrier(0);
            if (more && !(p.abl & 1)) issue_piece(k + 2, k & 1, piece);
            ++piece;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (dma) {                                       // NJ = 3: six MFMAs, seven pieces
#pragma unroll
        for (; piece < LPS; ++piece)
          if (more && !(p.abl & 1)) issue_piece(k + 2, k & 1, piece);
      }
    };
    load(0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    load(1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(a0, b0, false);
    __builtin_amdgcn_sched_barrier(0);
    load(2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1, false);
    __builtin_amdgcn_sched_barrier(0);
    load(3, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0, false);
    __builtin_amdgcn_s_setprio(0);
    bg_wait_lgkm0();                                                    // the last sub's fragments are in registers
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                       // nobody reads this stage any more
    __builtin_amdgcn_s_setprio(1);
    mma(a1, b1, true);
    __builtin_amdgcn_s_setprio(0);
  }

  // ---- epilogue: bias / scale / ReLU -> f16 -> LDS row-major (16-byte skew per row), then whole row segments
  const float lo = p.relu ? 0.f : -INFINITY;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int nc = n0 + wn * (32 * NJ) + j * 32;               // first column of this 32-column block
    const float sc = nc < p.scale_cols ? p.scale : 1.f;        // scale_cols is a multiple of 32
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + nc + 8 * g + 4 * lh);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + (lane & 31);
        const h4 hv = {(half_t)fmaxf((acc[i][j][4 * g + 0] + b4.x) * sc, lo), (half_t)fmaxf((acc[i][j][4 * g + 1] + b4.y) * sc, lo),
                       (half_t)fmaxf((acc[i][j][4 * g + 2] + b4.z) * sc, lo), (half_t)fmaxf((acc[i][j][4 * g + 3] + b4.w) * sc, lo)};
        *reinterpret_cast<h4a*>(smem + (size_t)row * XROW + (wn * (32 * NJ) + j * 32 + 8 * g + 4 * lh) * 2) = hv;
      }
    }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;                                   // 16-byte chunks per tile row
  if (p.blocked) {
    // blocked activation layout (kernels.h): 32 rows x 8 columns = 512 contiguous bytes; wave w writes row block w
    char* ob = reinterpret_cast<char*>(p.out) + ((size_t)((m0 >> 5) + wave) * (size_t)(p.N >> 3) + (size_t)(n0 >> 3)) * 512;
    const int r = lane & 31;
#pragma unroll 4
    for (int it = 0; it < CPR / 2; ++it) {
      const int cg = 2 * it + lh;
      const h8 v = *reinterpret_cast<const h8a*>(smem + (size_t)(wave * 32 + r) * XROW + cg * 16);
      if (!(p.abl & 4)) bg_store16(ob + (size_t)cg * 512 + r * 16, v);
    }
  } else {
    // row-major: wave w owns rows 32w .. 32w+31; consecutive lanes = consecutive 16-byte chunks of a row
#pragma unroll 4
    for (int it = 0; it < 32 * CPR / 64; ++it) {
      const int idx = it * 64 + lane;
      const int rl = idx / CPR, ch = idx - rl * CPR;
      const int row = wave * 32 + rl;
      const h8 v = *reinterpret_cast<const h8a*>(smem + (size_t)row * XROW + ch * 16);
      if (!(p.abl & 4)) bg_store16(p.out + (size_t)(m0 + row) * p.ldc + n0 + ch * 8, v);
    }
  }
}

bool gemm_big_applicable(const GemmArgs& a, int cus, int* nj_out) {
  if (!a.out_f16 || a.out_f32 || a.resid || a.add2 || !a.out_padded || a.a_blocked) return false;
  if (a.K % 64 != 0 || a.lda % 8 != 0 || a.ldw % 8 != 0 || a.scale_cols % 32 != 0) return false;
  if (!a.out_blocked && a.ldc16 % 8 != 0) return false;
  if (a.out_blocked && a.N % 64 != 0) return false;
  const int tm = cdiv(a.M, BG_BM);
  int best = 0;
  double best_cost = 0;
  for (int nj = 4; nj >= 3; --nj) {
    const int bn = 64 * nj;
    if (a.N % bn != 0) continue;
    const int tiles = tm * (a.N / bn);
    if (tiles < cus) continue;                         // fewer tiles than CUs: the 128-row tiles of gemm_f16_pp3 fill the chip better
    const double cost = (double)cdiv(tiles, cus) * bn; // rounds x tile width
    if (!best || cost < best_cost) { best = nj; best_cost = cost; }
  }
  if (!best) return false;
  *nj_out = best;
  return true;
}

void launch_gemm_big(hipStream_t s, const GemmArgs& a, int nj) {
  BigDev d;
  d.A = a.A; d.W = a.W; d.bias = a.bias; d.out = a.out_f16;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc = a.ldc16;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.tiles_m = cdiv(a.M, BG_BM); d.tiles_n = a.N / (64 * nj);
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.blocked = a.out_blocked;
  { static int abl = -1; if (abl < 0) { const char* e = getenv("PF_BIG_ABL"); abl = e ? atoi(e) : 0; } d.abl = abl; }
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bg_lds(4)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, bg_lds(3)));
      attr_set[dev & 63] = true;
    }
  }
  const dim3 grid((unsigned)(d.tiles_m * d.tiles_n));
  if (nj == 4) hipLaunchKernelGGL(gemm_big_kernel<4>, grid, dim3(512), bg_lds(4), s, d);
  else hipLaunchKernelGGL(gemm_big_kernel<3>, grid, dim3(512), bg_lds(3), s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
