// k_gemm_big.hip — persistent 256 x 256 tile GEMM for the FFN up-projection, f16 result in the blocked layout:
//
//   C[M,N] = relu(A[M,K] * W[N,K]^T + bias)  [* scale on the first columns]  -> blocked f16 (kernels.h)
//
// (the MatMul + Add + Relu nodes of the graph InferenceSession.Run executes, AliParaformerAsr/OfflineProjOfParaformer.cs:68).
//
// Why a second kernel next to gemm_f16_pp3 (k_gemm.hip): the K loop of both kernels is bound by the per-CU operand
// path, ~60 GB/s = 25 B/clk through LDS-DMA whatever the source (profiles/round2_gemm_cold_vs_warm.txt: a k-step takes
// time proportional to its operand bytes — 0.79 us for 48 KB, 0.535 us for 32 KB — on an otherwise idle chip with the
// operands L2-resident).  A 256 x 128 tile moves 48 KB per 4.2 MFLOP (87 flop/B, ceiling ~5.3 TFLOP/s per CU), a
// 256 x 256 tile 64 KB per 8.4 MFLOP (131 flop/B, ceiling ~7.9), so the wide projections want the big tile.  k-steps of
// 32, a 4-stage ring of 32 KB stages, THREE stages in flight, one s_barrier per k-step placed in the middle of the
// step: at mid-step k every wave has its stage-(k+1) pieces landed (counted vmcnt) and all its reads of stage k
// retired, so after the barrier (a) the first fragments of stage k+1 are read under the second half of step k's MFMAs
// and (b) slot k % 4 is free for the DMA of stage k+4, whose pieces go out between the MFMAs of the next half-steps.
// 8 waves as 4 x 2, wave tile 64 x 128 = 2 x 4 MFMA 32x32x16 blocks (128 accumulator registers).  64-byte LDS rows are
// swizzled chunk ^ ((row >> 2) & 3) on the DMA source and on the ds_read_b128 address (conflict-free for the 16-lane
// read groups).  Workgroups are dealt to XCDs in contiguous runs with n fastest, so an A panel is shared in one private
// L2.  (Rounds 1-2 also carried a one-tile-per-workgroup 256 x {192, 256} form with an LDS-transposed row-major
// epilogue; it measured equal to gemm_f16_pp3 on every shape — its per-tile prologue / epilogue is not overlapped at
// K = 512 — and was removed in round 3: profiles/round2_gemm_big_ablation.txt keeps the numbers.)
#include "kernels.h"

#include <algorithm>
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
};

// compile-time ablation bits for tools/abl_big.sh (timing experiments only; 0 in every shipped build — run-time flags
// put a branch and a full lgkmcnt(0) in front of every MFMA): 1 no DMA, 2 no MFMA, 4 no result stores, 8 no fragment
// reads, 16 no barriers (persistent form)
#ifndef PF_BIG_ABL
#define PF_BIG_ABL 0
#endif
constexpr int BG_ABL = PF_BIG_ABL;
constexpr int BG_BM = 256, BG_BK = 32, BG_ROWB = BG_BK * 2, BG_S = 4;
constexpr int bg_stage(int nj) { return (BG_BM + 64 * nj) * BG_ROWB; }          // A tile + W tile of one k-step
constexpr int bg_xrow(int nj) { return 64 * nj * 2 + 16; }                       // f16 epilogue row + 16-byte skew
constexpr int bg_lds(int nj) { return BG_S * bg_stage(nj) > BG_BM * bg_xrow(nj) ? BG_S * bg_stage(nj) : BG_BM * bg_xrow(nj); }

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

// ---------------------------------------------------------------------------------------------------------------
// Persistent form for the blocked-layout result (FFN-up: [M x 512] x [512 x 2048] + bias + ReLU -> the blocked f16
// hidden): one workgroup per CU walks tiles slot, slot + G, .. with the (tile, k-step) sequence as ONE flat pipeline,
// so a tile's prologue (three stages of latency) and the next workgroup's launch disappear behind the previous tile's
// k-steps.  A D^T fragment quad pair IS a 32-row x 8-column block of the blocked layout, so the tile end needs no LDS
// exchange: bias (from a 1 KiB LDS line fetched with the tile's first stage by a 4-byte LDS-DMA per lane) + scale +
// ReLU + cvt, then 32 fire-and-forget 512-byte stores per wave straight from the accumulators, which are zeroed and
// reused at once.  Those stores enter the vmcnt immediates exactly (vmcnt retires in order): the two waits that follow
// a tile end allow 32 more operations in flight.
__device__ __forceinline__ void bg_store8(void* p, h4 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bg_glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
constexpr int BGP_NJ = 4, BGP_BN = 64 * BGP_NJ, BGP_STAGE = bg_stage(BGP_NJ), BGP_RING = BG_S * BGP_STAGE;
constexpr int BGP_LDS = BGP_RING + 2 * BGP_BN * 4;            // ring + two bias lines

// (A variant with the two wave groups — waves 0-3 / 4-7, one of each per SIMD — running the same sequence half a k-step
// apart, two barriers and two landing waits per step, was built and measured: correct, 265 vs 199 us at M = 64000.)
__global__ __launch_bounds__(512, 1) void gemm_bigp_kernel(BigDev p) {
  constexpr int NJ = BGP_NJ, BN = BGP_BN, A_BYTES = BG_BM * BG_ROWB, STAGE = BGP_STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lh = lane >> 5;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 2) & 3; };

  // ---- tile schedule: persistent block b (on XCD b % 8) takes tiles slot, slot + G, ...; XCDs get contiguous runs
  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = p.K / BG_BK;
  const int T = n_my * nk;
  if (T == 0) return;
  float* const bias_line = reinterpret_cast<float*>(smem + BGP_RING);

  // ---- DMA cursor (uniform): stage is_t = k-step is_k of tile is_tile
  const int srow = lane >> 2, schunk = lane & 3;
  unsigned a_vo[2], w_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 8 * i) * 16 + srow;
    a_vo[i] = (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
    w_vo[i] = (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  int is_tile = slot, is_k = 0, is_t = 0, is_slot = 0, is_round = 0;
  const char* is_a;
  const char* is_w;
  int is_n0 = 0;
  auto set_issue_tile = [&]() __attribute__((always_inline)) {
    const int tm = is_tile / p.tiles_n, tn = is_tile - tm * p.tiles_n;
    is_a = reinterpret_cast<const char*>(p.A + (size_t)tm * BG_BM * p.lda);
    is_w = reinterpret_cast<const char*>(p.W + (size_t)tn * BN * p.ldw);
    is_n0 = tn * BN;
  };
  set_issue_tile();
  auto issue_piece = [&](int q) __attribute__((always_inline)) {
    if (BG_ABL & 1) return;
    char* st = smem + (is_slot & (BG_S - 1)) * STAGE + wave * 1024;
    if (q < 2) bg_glds16(is_a + a_vo[q & 1], st + (q & 1) * 8192);
    else bg_glds16(is_w + w_vo[q & 1], st + A_BYTES + (q & 1) * 8192);
  };
  // The loop body is BRANCH-FREE (a conditional DMA or wait splits the block, and the compiler then puts a full
  // lgkmcnt(0) in front of the first MFMA: the fragment reads issued just before it would no longer run under the
  // MFMAs).  So every step issues exactly five operations per wave — the bias columns of the cursor's tile -> the bias
  // line of that round's parity (4 bytes per lane; waves 4-7 repeat waves 0-3), then the four pieces of the cursor's
  // stage — and past the end of the tile list the cursor simply stops advancing (the last stage is re-loaded into a
  // slot nobody reads again), which makes every vmcnt immediate a compile-time constant.
  auto issue_bias = [&]() __attribute__((always_inline)) {
    if (BG_ABL & 1) return;
    bg_glds4(p.bias + is_n0 + (wave & 3) * 64 + lane, bias_line + (is_round & 1) * BN + (wave & 3) * 64);
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    if (is_t + 1 < T) {
      ++is_t;
      is_a += BG_BK * 2; is_w += BG_BK * 2;
      if (++is_k == nk) { is_k = 0; is_tile += G; ++is_round; set_issue_tile(); }
    }
    ++is_slot;
  };
  // stage t + 1 of this wave has landed: the 2 x 5 operations of stages t + 2, t + 3 (and, for two steps after a tile
  // end, its 32 stores) may still be in flight
  auto wait_landed = [&](bool burst) __attribute__((always_inline)) {
    if (burst) bg_wait_vmcnt<42>(); else bg_wait_vmcnt<10>();
  };

  unsigned fa[2][2], fb[2][NJ];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
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
  h8 a0[2] = {}, b0[NJ] = {}, a1[2] = {}, b1[NJ] = {};
  // Fragment reads and their waits are inline asm: the compiler's own wait insertion loses count across the loop
  // back-edge (a full lgkmcnt(0) in front of the first MFMA, so the reads of the second half never ran under the first
  // half's MFMAs) and answers LDS reads behind in-flight LDS-DMA with vmcnt(0) (at the tile end that drained the three
  // stages in flight AND the 32 result stores: 5-6 us per tile).  `rd` is the stage's byte offset in LDS.
  auto load = [&](unsigned rd, int s, h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    if (BG_ABL & 8) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bf[j]) : "v"(rd + fb[s][j]) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[i]) : "v"(rd + fa[s][i]) : "memory");
  };
  // wait until at most N LDS reads are outstanding; the fragments are operands so that no use moves above the wait
  auto frag_wait6 = [&](h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
  };
  auto frag_wait0 = [&](h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
  };
  // 2*NJ MFMAs of one sub-step; the pieces [q0, q1) of stage kd go out between them (kd < 0: none)
  auto mma = [&](h8 (&af)[2], h8 (&bf)[NJ], bool dma) __attribute__((always_inline)) {
    int q = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (!(BG_ABL & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (q < 4) {
          __builtin_amdgcn_sched_barrier(0);
          if (dma) issue_piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };

  // ---- prologue: three stages in flight
#pragma unroll
  for (int st = 0; st < BG_S - 1; ++st) {
    issue_bias();
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(q);
    issue_advance();
  }
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring
  wait_landed(false);                                          // stage 0 (and the first bias line) of this wave
  __builtin_amdgcn_s_barrier();
  load(lds0, 0, a0, b0);

  int tile = slot, k = 0, round = 0, since_burst = 3;
  // ---- tile end: bias + scale + ReLU + cvt, 2 * NJ * 4 = 32 blocked-layout stores per wave, accumulators reused at once
  // accumulators START as the bias of their tile (a 16-byte LDS read per register quad), so the tile end is only
  // [scale] -> cvt -> packed ReLU -> store, and the next tile's bias is loaded in place of the zeroing
  typedef float f4v __attribute__((ext_vector_type(4)));
  auto acc_init = [&](int rnd) __attribute__((always_inline)) {
    const unsigned bl = lds0 + BGP_RING + ((rnd & 1) * BN + wn * (32 * NJ) + 4 * lh) * 4;
    f4v b4[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(b4[j][g]) : "v"(bl + (j * 32 + 8 * g) * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b4[0][0]), "+v"(b4[0][1]), "+v"(b4[0][2]), "+v"(b4[0][3]), "+v"(b4[1][0]), "+v"(b4[1][1]), "+v"(b4[1][2]), "+v"(b4[1][3]),
                   "+v"(b4[2][0]), "+v"(b4[2][1]), "+v"(b4[2][2]), "+v"(b4[2][3]), "+v"(b4[3][0]), "+v"(b4[3][1]), "+v"(b4[3][2]), "+v"(b4[3][3]));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j][4 * g + 0] = b4[j][g][0]; acc[i][j][4 * g + 1] = b4[j][g][1]; acc[i][j][4 * g + 2] = b4[j][g][2]; acc[i][j][4 * g + 3] = b4[j][g][3];
        }
  };
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  auto tile_end = [&]() __attribute__((always_inline)) {
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BG_BM, n0 = tn * BN;
    char* ob = reinterpret_cast<char*>(p.out) + ((size_t)((m0 >> 5) + wm * 2) * (size_t)(p.N >> 3) + (size_t)((n0 + wn * (32 * NJ)) >> 3)) * 512 +
               (lane & 31) * 16 + lh * 8;
    const size_t rb_stride = (size_t)(p.N >> 3) * 512;
    const h2v zero2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nc = n0 + wn * (32 * NJ) + j * 32;
      const bool scaled = nc < p.scale_cols;                   // uniform; scale_cols is a multiple of 32
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f2v lo2 = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]}, hi2 = {acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if (scaled) { lo2 *= p.scale; hi2 *= p.scale; }
          h2v l = __builtin_convertvector(lo2, h2v), h = __builtin_convertvector(hi2, h2v);
          if (p.relu) { l = __builtin_elementwise_max(l, zero2); h = __builtin_elementwise_max(h, zero2); }
          const h4 hv = {l[0], l[1], h[0], h[1]};
          if (!(BG_ABL & 4)) bg_store8(ob + i * rb_stride + (size_t)(j * 4 + g) * 512, hv);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    k = 0; tile += G; ++round; since_burst = 0;
    acc_init(round);                                           // landed with the next tile's first stage, three steps ago
  };

  acc_init(0);                                                 // the first tile's bias (or zeros) landed before the barrier above
  for (int t = 0; t < T; ++t) {
    const unsigned rd = lds0 + (t & (BG_S - 1)) * STAGE;
    __builtin_amdgcn_sched_barrier(0);
    load(rd, 1, a1, b1);                                       // second half of stage t, under the first half's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    issue_bias();
    frag_wait6(a0, b0);                                        // first half (requested behind the last barrier) is in
    __builtin_amdgcn_s_setprio(1);
    mma(a0, b0, true);                                         // + the four pieces of stage t + 3 (slot freed by the barrier of step t - 1)
    __builtin_amdgcn_s_setprio(0);
    issue_advance();
    __builtin_amdgcn_sched_barrier(0);
    wait_landed(since_burst < 2 && !(BG_ABL & 4));             // stage t + 1 of this wave has landed
    frag_wait0(a1, b1);                                        // every read of stage t has retired
    __builtin_amdgcn_sched_barrier(0);
    if (!(BG_ABL & 16)) __builtin_amdgcn_s_barrier();
    load(lds0 + ((t + 1) & (BG_S - 1)) * STAGE, 0, a0, b0);    // (after the last step: a slot nobody uses)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(a1, b1, false);
    __builtin_amdgcn_s_setprio(0);
    ++since_burst;
    if (++k == nk) tile_end();
  }
  bg_wait_vmcnt<0>();                                          // the clamped DMA of the last steps must not outlive the workgroup's LDS
}

bool gemm_bigp_applicable(const GemmArgs& a) {
  if (!a.out_f16 || a.out_f32 || a.resid || a.add2 || !a.out_padded || a.a_blocked || !a.out_blocked) return false;
  if (a.K % 64 != 0 || a.K < 128 || a.lda % 8 != 0 || a.ldw % 8 != 0 || a.scale_cols % 32 != 0 || a.N % BGP_BN != 0 || a.N > 65536) return false;
  return true;                                               // K >= 128: a bias line is reused two tiles later
}

void launch_gemm_bigp(hipStream_t s, const GemmArgs& a, int cus) {
  BigDev d{};
  d.A = a.A; d.W = a.W; d.bias = a.bias; d.out = a.out_f16;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc = a.ldc16;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.tiles_m = cdiv(a.M, BG_BM); d.tiles_n = a.N / BGP_BN;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.blocked = 1;
  if (!d.bias) {                                             // the kernel always loads a bias line: zeros when there is none
    static std::mutex zmu;
    static float* zeros[64] = {nullptr};
    int zd = 0;
    PF_HIP(hipGetDevice(&zd));
    std::lock_guard<std::mutex> lk(zmu);
    if (!zeros[zd & 63]) {
      PF_HIP(hipMalloc(&zeros[zd & 63], 65536 * 4));
      PF_HIP(hipMemset(zeros[zd & 63], 0, 65536 * 4));
      PF_HIP(hipDeviceSynchronize());                       // the fill runs on the null stream; the caller's stream does not wait for it
    }
    d.bias = zeros[zd & 63];
  }
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_bigp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BGP_LDS));
      attr_set[dev & 63] = true;
    }
  }
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  note_gemm_kernel("gemm_bigp_kernel");
  hipLaunchKernelGGL(gemm_bigp_kernel, dim3((unsigned)std::min(total, cus)), dim3(512), BGP_LDS, s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
