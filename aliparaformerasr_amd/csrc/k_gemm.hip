// k_gemm.hip — f16 x f16 -> f32 GEMM on the gfx950 matrix cores.
//
// Replaces the MatMul/Gemm nodes ONNX Runtime executes inside InferenceSession.Run
// (reference call site AliParaformerAsr/OfflineProjOfParaformer.cs:68): QKV, attention
// output, FFN, CIF conv (as im2col GEMM), decoder projections and the vocabulary layer.
//
//   C[M,N] = A[M,K] * W[N,K]^T   (both operands K-contiguous, "B^T input")
//
// Structure (every choice below was driven by PMC counters / an in-kernel s_memtime timeline,
// see DESIGN.md §4.1 and the git history of this file):
//  * PERSISTENT: one 512-thread workgroup per CU walks a static tile list; the (tile, k-step)
//    sequence is ONE flat software pipeline (the model's GEMMs have K = 512: 8 k-steps per
//    tile, a per-tile prologue would dominate).  Tiles are dealt so that each of the 8 XCDs
//    works on a contiguous run (n fastest): A panels / W tiles are shared in its private L2.
//  * Tile 256x128x64; 8 wavefronts, each a 64x64 block = 2x2 v_mfma_f32_32x32x16_f16.
//  * PING-PONG: waves 0-3 (group A) and 4-7 (group B) share the 4 SIMDs pairwise; while one
//    group issues its 16-MFMA burst for k-step k the other reads its fragments for the same
//    step from LDS.  One s_barrier per phase, placed INSIDE the burst (after 12 of 16 MFMAs)
//    so the matrix pipe does not drain at the phase boundary.
//      phase 2k   : A = LOAD(k)                    B = COMPUTE(k-1) + DMA(k+S-1)
//      phase 2k+1 : A = COMPUTE(k) + DMA(k+S-1)    B = LOAD(k)
//  * Operands go HBM -> LDS with global_load_lds_dwordx4 into an S-stage ring, consumed after
//    a COUNTED s_waitcnt vmcnt(N): S-1 stages stay in flight across barriers.  The 6 DMA pieces
//    of a wave are slotted between its MFMAs (a burst of 48 at a phase boundary stalls all
//    waves on the CU's address unit).  Past the end of the tile list the DMA is clamped (it
//    re-loads the last step into a ring slot nobody reads again), so every wait immediate is a
//    compile-time constant and the steady state has no tail branches.  The bank-conflict XOR
//    swizzle is applied to the DMA SOURCE address and to the ds_read_b128 address.
//  * The MFMA is issued as D^T = W_tile * X_tile^T: a lane owns one output row and 4 consecutive
//    columns per register quad.
//  * f16 results (QKV, FFN-up, decoder q / kv): the accumulators of a tile START as the bias
//    (read from a 256-byte LDS line that is itself fetched a tile ahead by LDS-DMA), so at tile
//    end the conversion is only [q-scale] [ReLU] cvt_pk into 32 packed registers and the
//    accumulators are free again at once.  The packed tile is then transposed through a
//    2 KiB/wave LDS scratch 8 rows at a time, one pass per following k-step (slotted into the wave's own burst, under the
//    partner group's MFMAs), and stored as whole 128-byte lines after that step's own MFMAs.
//    Output rows are padded (GemmArgs::out_padded) so no store is ever predicated off and the
//    store count enters the vmcnt immediates exactly (vmcnt is in-order on gfx9).
//    (Double-buffered accumulators were tried instead: 128 + 64 fragment VGPRs spill.)
//  * fp32 results (+ residual, + FSMN add): at tile end, 4 rows at a time through the same scratch so that
//    residual loads and stores are whole row segments (lds_epilogue32); edge tiles use direct 16-byte I/O.
//  * Two kernel kinds (f16-only / fp32) x two tile heights (256 / 128 rows, MI = 2 / 1) are instantiated.
#include "kernels.h"
#include "exact.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
// LDS scratch is written with one vector width and read back with another: these accesses must
// not be reordered by type-based alias analysis
typedef h8 __attribute__((may_alias)) h8a;
typedef h4 __attribute__((may_alias)) h4a;
typedef float4 __attribute__((may_alias)) float4a;
template <int V>
using ic = std::integral_constant<int, V>;

struct GemmDev {
  const half_t* A; const half_t* W; const float* bias;
  float* out_f32; half_t* out_f16; const float* resid; const float* add2;
  int lda, ldw, ldc32, ldc16, ldr, ld2;
  int M, N, K, tiles_m, tiles_n;
  int relu, scale_cols; float scale;
  int out_padded;
  int out_blocked, a_blocked;
  int f16_lo_off;            // > 0: out_f16 receives the result as an x3 pair: hi at column n, lo' = f16((v - hi) * 2^11) at n + f16_lo_off
  int k_wrap, a_wrap, w_wrap;    // K-loop wrap (kernels.h): k-step index, cursor steps back in BYTES; 0 = none
  float wrap_scale;
  // int8 variant (gemm_i8_pp3): A / W hold SIGNED bytes a' = a_q - 128, w' = w_q - 128 (K counted in bytes);
  // acc = sum a' w' is corrected to sum (a_q - a_zp)(w_q - w_zp[n]) with the row / column sums and dequantised
  const int32_t* q_rowsum;   // [M]  sum_k a'[m,k]
  const int32_t* q_colsum;   // [N]  sum_k w'[n,k]
  const int32_t* q_wzp;      // [N]  w_zp[n] - 128
  const float* q_wscale;     // [N]
  const float* q_aparams;    // [2]  {a_scale, a_zp} of the dynamically quantised activation tensor (device, written by the quantise kernels)
  int q_k;                   // true K (before padding)
  // f16-result int8 kernel (gemm_i8f_pp3): per column dz[n] = (colsum[n] - K w_zp'[n]) * 256 + (w_zp'[n] & 255), so that a
  // tile's column constants are three 256-byte lines (dz, w_scale, bias) fetched by LDS-DMA a tile ahead
  const int32_t* q_dz;       // [N]
  float* q_part;             // null, or QMM_G {min, max} pairs: the result's range for the quantiser that consumes it (k_quant.hip)
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // gfx9 s_waitcnt simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// compile-time ablation bits for tools/gemm_abl.sh (timing experiments only; 0 in every shipped build):
// 1 no steady-state DMA, 2 no fragment reads, 4 no MFMA, 8 no f16 output stores, 16 no direct (fp32) epilogue, 32 no transpose passes
#ifndef PF_ABL
#define PF_ABL 0
#endif
constexpr int ABL = PF_ABL;
// cache policy of the deferred f16 result stores: 0 plain, 1 nt, 2 sc1 (write-through, line dropped from the
// XCD's L2).  66 MB of results per launch otherwise churn the 8 x 4 MB L2s that hold the A panels and W tiles.
// A/B in one session (tools/gemm_st.sh): plain 14.88-14.96 ms/step, nt 14.96-14.99 (the consumers then miss),
// sc1 14.81 (QKV -5 %, FFN-up -3 %, attention / FSMN / FFN-down unchanged).
#ifndef PF_GEMM_ST
#define PF_GEMM_ST 2
#endif
__device__ __forceinline__ void st16(void* p, h8 v) {
#if PF_GEMM_ST == 1
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
#elif PF_GEMM_ST == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<h8*>(p) = v;
#endif
}
__device__ __forceinline__ void st8(void* p, h4 v) {
#if PF_GEMM_ST == 1
  asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
#elif PF_GEMM_ST == 2
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<h4*>(p) = v;
#endif
}
constexpr int GEMM_BN = 128, GEMM_BK = 64, GEMM_S = 3;   // tile rows: 128 * MI (MI = 32-row MFMA blocks per wave)
constexpr int GEMM_SCRATCH = 8 * 2048;                                   // 2 KiB per wave
constexpr int gemm_lds_bytes(int mi) { return GEMM_S * (128 * mi + GEMM_BN) * GEMM_BK * 2 + GEMM_SCRATCH; }   // MI=2: 160 KiB

// KIND 1: f16-only results (deferred packed epilogue; direct epilogue only for wave tiles straddling N)
// KIND 2: fp32 results / residual / FSMN add (LDS row-segment epilogue; direct epilogue for edge tiles)
// KIND 3: f16-only results in the blocked activation layout (kernels.h): fragments stored as whole lines, no transposition
// MI: 32-row MFMA blocks per wave (2 -> 256-row tiles; 1 -> 128-row tiles for GEMMs whose 256-row
//     tile count would leave most CUs idle, i.e. the decoder's M = B*L rows)
typedef int i16x __attribute__((ext_vector_type(16)));
typedef int i4x __attribute__((ext_vector_type(4)));
typedef i4x __attribute__((may_alias)) i4xa;

// I8: the same pipeline on v_mfma_i32_32x32x32_i8 — a k-step is still 128 bytes per row (128 int8 instead of 64 f16),
// fragments are still 16 bytes per lane, so staging, swizzle and fragment reads are unchanged; the accumulators are
// int32 and the (KIND 2) epilogues dequantise: y = float(acc - corr) * (a_scale * w_scale[n]) + bias[n], exact integers.
template <int KIND, int MI, bool I8>
__device__ __forceinline__ void gemm_pp3_impl(const GemmDev& p) {
  static_assert(!I8 || KIND == 2 || KIND == 1, "the int8 variant has the fp32-result epilogues and the row-major f16 one");
  using acc_t = std::conditional_t<I8, i16x, f16x>;
  constexpr int BK = GEMM_BK, S = GEMM_S, BM = 128 * MI, BN = GEMM_BN, WM = 32 * MI;
  constexpr int WN = 2, NW = 8;
  constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPI = 64 / CPR;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NW, B_PW = B_BYTES / 1024 / NW;
  constexpr int LPS = A_PW + B_PW;                   // LDS-DMA pieces per wave per k-step
  constexpr int KSUB = BK / 16;
  constexpr int WAITN = LPS * (S - 2);               // DMA pieces allowed in flight at a wait
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                         // 0 = A (tile rows 0..127), 1 = B (128..255)
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5;

  // ---- tile schedule: persistent block b (on XCD b % 8) takes tiles slot, slot+G, ...
  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = I8 ? p.K / (2 * BK) : p.K / BK;      // int8: K is in bytes, a k-step is 128 of them
  const int T = n_my * nk;
  if (T == 0) return;

  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 1) & 7; };      // 128-byte rows: chunk ^ ((row>>1)&7)

  // ---- LDS-DMA addressing, hoisted: per-lane constant byte offsets + a uniform base pointer
  const int srow = lane / CPR, schunk = lane % CPR;
  unsigned a_vo[A_PW], w_vo[B_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = (wave + NW * i) * RPI + srow;
    a_vo[i] = I8 ? (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 4)) : (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
    if (p.a_blocked) {                               // piece pi = 1 KiB = 2 column groups of one 32-row block
      const int pi = wave + NW * i;
      a_vo[i] = (unsigned)((pi >> 2) * (p.K >> 3) * 512 + (pi & 3) * 1024 + lane * 16);
    }
  }
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int row = (wave + NW * i) * RPI + srow;
    w_vo[i] = I8 ? (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 4)) : (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  int is_tile = slot, is_kt = 0, is_left = T;        // DMA cursor (uniform); clamps at the last step
  char* is_lds = smem + wave * 1024;
  const char* is_a;
  const char* is_w;
  auto set_issue_tile = [&]() __attribute__((always_inline)) {
    const int tm = is_tile / p.tiles_n, tn = is_tile - tm * p.tiles_n;
    is_a = p.a_blocked ? reinterpret_cast<const char*>(p.A) + (size_t)tm * (BM / 32) * (size_t)(p.K >> 3) * 512
                       : reinterpret_cast<const char*>(p.A) + (size_t)tm * BM * p.lda * (I8 ? 1 : 2);
    is_w = reinterpret_cast<const char*>(p.W) + (size_t)tn * BN * p.ldw * (I8 ? 1 : 2);
  };
  set_issue_tile();
  auto issue_piece = [&](int i) __attribute__((always_inline)) {
    if constexpr (ABL & 1) return;
    if (i < A_PW) glds16(is_a + a_vo[i < A_PW ? i : 0], is_lds + NW * i * 1024);
    else glds16(is_w + w_vo[i >= A_PW ? i - A_PW : 0], is_lds + A_BYTES + NW * (i - A_PW) * 1024);
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    is_lds = (is_lds + STAGE == smem + wave * 1024 + S * STAGE) ? smem + wave * 1024 : is_lds + STAGE;
    if (--is_left > 0) {
      is_a += p.a_blocked ? (BK / 8) * 512 : BK * 2; is_w += BK * 2;
      if (++is_kt == nk) { is_kt = 0; is_tile += G; set_issue_tile(); }
      else if (is_kt == p.k_wrap) { is_a -= p.a_wrap; is_w -= p.w_wrap; }      // (k_wrap = 0 never matches: is_kt >= 1 here)
    }
  };
  auto issue_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < LPS; ++i) issue_piece(i);
    issue_advance();
  };

  // ---- fragment read offsets inside a stage (bytes), hoisted
  unsigned fa[KSUB][MI], fb[KSUB][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * WM + i * 32 + (lane & 31);
    const int rb = wn * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
      if (i < MI)
        fa[s][i < MI ? i : 0] = p.a_blocked ? (unsigned)((((ra >> 5) * 8 + 2 * s + lh) * 32 + (ra & 31)) * 16)
                                            : (unsigned)(ra * ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
      fb[s][i] = (unsigned)(A_BYTES + rb * ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  }
  h8 af[KSUB][MI], bf[KSUB][2];
  const char* rd = smem;
  auto load_frags = [&]() __attribute__((always_inline)) {
    if constexpr (!(ABL & 2))
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < MI) af[s][i < MI ? i : 0] = *(const h8*)(rd + fa[s][i < MI ? i : 0]);
        bf[s][i] = *(const h8*)(rd + fb[s][i]);
      }
    rd = (rd + STAGE == smem + S * STAGE) ? smem : rd + STAGE;
  };

  // ---- accumulators + packed copy of the previous tile
  acc_t acc[MI][2];
  h4 hq[MI][2][4];                                      // finished tile, packed f16, awaiting its passes

  // ---- deferred f16 epilogue state
  char* const scr = smem + S * STAGE + wave * 2048;            // [0,1152) transpose rows, [1152,1408) bias line
  constexpr bool BLK = KIND == 3;
  const bool fast_kind = KIND == 1 || KIND == 3;   // launch_gemm checked: f16 only, no residual/add, padded rows, ldc16 % 8 == 0
  const float lo = p.relu ? 0.f : -INFINITY;
  char* const wp = scr + (lane & 7) * 144 + lh * 8;
  const char* const rp = scr + (lane >> 3) * 144 + (lane & 7) * 16;
  int pend = 0;                                      // passes of the finished tile still to emit
  char* ob = nullptr;                                // blocked layout: output pointer of the finished tile (this lane)
  half_t* op = nullptr;                              // its output pointer (row lane>>3, col (lane&7)*8)
  h8 rowv;
  half_t* rowp = nullptr;
  {                                                  // zero the bias line (bias == null -> stays zero)
    float* bl = reinterpret_cast<float*>(scr + (I8 ? 1664 : 1152));
    bl[lane] = 0.f;
    asm volatile("" ::: "memory");
  }
  // bias of tile `tile` -> the bias line, by a 4-byte LDS-DMA per lane (no VGPR, no compiler-inserted
  // wait; counted like any other DMA — ignoring it in the wait immediates only over-waits by one piece)
  auto fetch_bias = [&](int tile) __attribute__((always_inline)) {
    const int tn = tile - (tile / p.tiles_n) * p.tiles_n;
    int n = tn * BN + wn * 64 + lane;
    n = n < p.N ? n : p.N - 1;
    if constexpr (I8) {                              // int8: the tile's column constants, consumed at ITS end (three lines)
      glds4(p.q_dz + n, scr + 1152);
      glds4(p.q_wscale + n, scr + 1408);
      if (p.bias) glds4(p.bias + n, scr + 1664);
    } else {
      if (!p.bias) return;
      glds4(p.bias + n, scr + 1152);
    }
  };
  // accumulators of the next tile: bias (fast path: lane's 4 columns per quad, same for every row) or 0
  auto init_acc = [&](bool with_bias) __attribute__((always_inline)) {
    const float* bl = reinterpret_cast<const float*>(scr + 1152) + 4 * lh;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (!I8) { if (with_bias) b4 = *reinterpret_cast<const float4a*>(bl + j * 32 + 8 * g); }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][j][4 * g + 0] = b4.x; acc[i][j][4 * g + 1] = b4.y; acc[i][j][4 * g + 2] = b4.z; acc[i][j][4 * g + 3] = b4.w;
        }
      }
  };

  // one deferred pass: rows 8*idx .. 8*idx+7 of this wave's block (already packed in hq), split in
  // two halves that are slotted between the MFMAs of the wave's own COMPUTE phase (the matrix pipe is
  // busy 32 cycles per MFMA, the wave issues for 4): first the 16 owner lanes write the 8 rows to the
  // scratch, a few MFMAs later every lane reads one 16-byte row piece back; the store follows in `mid`.
  // (Doing this in the LOAD phase put ~400 cycles in front of the fragment reads of every phase.)
  auto pass_write = [&](int idx) __attribute__((always_inline)) {
    if constexpr (ABL & 32) return;
    const int q = idx & 3;
    const bool mine = ((lane & 31) >> 3) == q;
    auto body = [&](auto IC) __attribute__((always_inline)) {
      constexpr int i = decltype(IC)::value;
      if (mine) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<h4a*>(wp + (j * 32 + 8 * g) * 2) = hq[i][j][g];
      }
    };
    if (MI == 1 || idx < 4) body(ic<0>{});
    else body(ic<MI - 1>{});
    asm volatile("" ::: "memory");
  };
  auto pass_read = [&](int idx) __attribute__((always_inline)) {
    if constexpr (ABL & 32) return;
    asm volatile("" ::: "memory");
    rowv = *reinterpret_cast<const h8a*>(rp);
    asm volatile("" ::: "memory");
    rowp = op + (size_t)(idx * 8) * p.ldc16;
  };
  auto pass = [&](int idx) __attribute__((always_inline)) { pass_write(idx); pass_read(idx); };
  auto pass_store = [&]() __attribute__((always_inline)) { if constexpr (!(ABL & 8)) st16(rowp, rowv); };     // padded rows: always issued
  // blocked layout: "pass" idx = the two stores of (i = idx >> 2, g = idx & 3), j = 0, 1 — straight from hq
  auto blk_store = [&](int idx) __attribute__((always_inline)) {
    auto body = [&](auto IC, auto GC) __attribute__((always_inline)) {
      constexpr int i = decltype(IC)::value, g = decltype(GC)::value;
      char* o = ob + ((size_t)i * (p.N >> 3) + g) * 512;
      st8(o, hq[i][0][g]);
      st8(o + 4 * 512, hq[i][1][g]);
    };
    constexpr int I1 = MI - 1;
    switch (idx) {
      case 0: body(ic<0>{}, ic<0>{}); break;
      case 1: body(ic<0>{}, ic<1>{}); break;
      case 2: body(ic<0>{}, ic<2>{}); break;
      case 3: body(ic<0>{}, ic<3>{}); break;
      case 4: body(ic<I1>{}, ic<0>{}); break;
      case 5: body(ic<I1>{}, ic<1>{}); break;
      case 6: body(ic<I1>{}, ic<2>{}); break;
      default: body(ic<I1>{}, ic<3>{}); break;
    }
  };
  auto flush = [&]() __attribute__((always_inline)) {
    while (pend > 0) {
      if constexpr (BLK) { blk_store(4 * MI - pend); --pend; }
      else { pass(4 * MI - pend); --pend; pass_store(); }
    }
  };

  // int8: exact integer correction + dequantisation of one accumulator value (row m, column n)
  //   sum (a_q - a_zp)(w_q - w_zp) = acc - w_zp' rowsum[m] - a_zp' colsum[n] + K a_zp' w_zp'   (primes: minus 128)
  //   y = float(that) * (a_scale * w_scale[n])            [two roundings; bias / residual follow as separate adds]
  float q_ascale = 1.f;
  int q_azp = 0;
  if constexpr (I8) { q_ascale = p.q_aparams[0]; q_azp = (int)p.q_aparams[1] - 128; }
  float trk_lo = 0.f, trk_hi = 0.f;                   // range of the f16 results (p.q_part): the consumer's DynamicQuantizeLinear pass 1
  auto dequant = [&](int a, int rowsum, int n) __attribute__((always_inline)) -> float {
    const int wz = p.q_wzp[n];
    const int v = a - wz * rowsum - q_azp * p.q_colsum[n] + p.q_k * q_azp * wz;
    return mul_rn((float)v, mul_rn(q_ascale, p.q_wscale[n]));
  };

  // direct epilogue (fp32 results, residual / FSMN add, or a wave tile that straddles N)
  auto direct_epilogue = [&](int tile) __attribute__((always_inline)) {
    if constexpr (ABL & 16) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM + wm * WM, n0 = tn * BN + wn * 64;
    const bool interior = (m0 + WM <= p.M) && (n0 + 64 <= p.N);
    const int nb = n0 + 4 * lh;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + i * 32 + (lane & 31);
      const float* r_ptr = p.resid ? p.resid + (size_t)m * p.ldr + nb : nullptr;
      const float* a_ptr = p.add2 ? p.add2 + (size_t)m * p.ld2 + nb : nullptr;
      float* o32 = p.out_f32 ? p.out_f32 + (size_t)m * p.ldc32 + nb : nullptr;
      half_t* o16 = p.out_f16 ? p.out_f16 + (size_t)m * p.ldc16 + nb : nullptr;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dn = j * 32 + 8 * g;
          const int n = nb + dn;
          float v[4];
          if constexpr (I8) {
            const int rs = m < p.M ? p.q_rowsum[m] : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = n + e < p.N ? dequant(acc[i][j][4 * g + e], rs, n + e) : 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          }
          acc[i][j][4 * g + 0] = 0; acc[i][j][4 * g + 1] = 0; acc[i][j][4 * g + 2] = 0; acc[i][j][4 * g + 3] = 0;
          if (!interior && (m >= p.M || n >= p.N)) continue;
          if (interior || n + 3 < p.N) {
            if (p.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (n < p.scale_cols) { v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale; }
            if (a_ptr) {
              const float4 a4 = *reinterpret_cast<const float4*>(a_ptr + dn);
              v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
            }
            if (r_ptr) {
              const float4 r4 = *reinterpret_cast<const float4*>(r_ptr + dn);
              v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = fmaxf(v[e], lo);
              if constexpr (I8) { trk_lo = fminf(trk_lo, v[e]); trk_hi = fmaxf(trk_hi, v[e]); }
            }
            if (o32) *reinterpret_cast<float4*>(o32 + dn) = make_float4(v[0], v[1], v[2], v[3]);
            if (o16) {
              const h4 hv = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              *reinterpret_cast<h4*>(o16 + dn) = hv;
              if (p.f16_lo_off > 0)      // math_mode 3: the next product's operand pair, written where it is produced
                *reinterpret_cast<h4*>(o16 + dn + p.f16_lo_off) =
                    h4{(half_t)((v[0] - (float)hv[0]) * 2048.f), (half_t)((v[1] - (float)hv[1]) * 2048.f),
                       (half_t)((v[2] - (float)hv[2]) * 2048.f), (half_t)((v[3] - (float)hv[3]) * 2048.f)};
            }
          } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) {
              float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
              if (n + e < p.scale_cols) x *= p.scale;
              if (a_ptr) x += a_ptr[dn + e];
              if (r_ptr) x += r_ptr[dn + e];
              x = fmaxf(x, lo);
              if constexpr (I8) { trk_lo = fminf(trk_lo, x); trk_hi = fmaxf(trk_hi, x); }
              if (o32) o32[dn + e] = x;
              if (o16) {
                o16[dn + e] = (half_t)x;
                if (p.f16_lo_off > 0) o16[dn + e + p.f16_lo_off] = (half_t)((x - (float)(half_t)x) * 2048.f);
              }
            }
          }
        }
      }
    }
  };

  // fp32 epilogue through the wave's LDS scratch (interior wave tiles of fp32 results): the D^T
  // fragment gives a lane one output ROW, so direct 16-byte accesses touch 32 different rows per
  // instruction (32 bytes used of every 128-byte line; ~4x the address-unit cycles of a full-line
  // access, and the CU's address unit is what the residual-carrying GEMMs wait on).  Instead 4 rows
  // at a time are written to the scratch by their 8 owner lanes and read back row-contiguous
  // (16 lanes x 16 B = one 256-byte row segment), so residual / FSMN-add loads and the stores are
  // whole lines.
  auto lds_epilogue32 = [&](int tile) __attribute__((always_inline)) {
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM + wm * WM, n0 = tn * BN + wn * 64;
    const int lc = lane & 31;
    const int rr = lane >> 4, cc = lane & 15;              // reader: row in chunk, 4-column group
    const int n = n0 + cc * 4;
    char* const wq = scr + (lc & 3) * 272 + lh * 16;
    const char* const rq = scr + rr * 272 + cc * 16;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    const float sc = (n < p.scale_cols) ? p.scale : 1.f;
    // int8: the four columns' correction terms and scales, fixed for this lane
    i4x q_wz = {0, 0, 0, 0}, q_cs = {0, 0, 0, 0};
    float4 q_sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (I8) {
      q_wz = *reinterpret_cast<const i4x*>(p.q_wzp + n);
      q_cs = *reinterpret_cast<const i4x*>(p.q_colsum + n);
      const float4 ws = *reinterpret_cast<const float4*>(p.q_wscale + n);
      q_sc = make_float4(mul_rn(q_ascale, ws.x), mul_rn(q_ascale, ws.y), mul_rn(q_ascale, ws.z), mul_rn(q_ascale, ws.w));
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int m = m0 + i * 32 + c * 4 + rr;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f), a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.resid) r4 = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
        if (p.add2) a4 = *reinterpret_cast<const float4*>(p.add2 + (size_t)m * p.ld2 + n);
        int q_rs = 0;
        if constexpr (I8) q_rs = p.q_rowsum[m];
        if ((lc >> 2) == c) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if constexpr (I8)
                *reinterpret_cast<i4xa*>(wq + (j * 32 + 8 * g) * 4) =
                    i4x{acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
              else
                *reinterpret_cast<float4a*>(wq + (j * 32 + 8 * g) * 4) =
                    make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
        }
        asm volatile("" ::: "memory");
        float4 v;
        if constexpr (I8) {
          const i4x a = *reinterpret_cast<const i4xa*>(rq);
          const int kz = p.q_k * q_azp;
          v.x = mul_rn((float)(a[0] - q_wz[0] * q_rs - q_azp * q_cs[0] + kz * q_wz[0]), q_sc.x);
          v.y = mul_rn((float)(a[1] - q_wz[1] * q_rs - q_azp * q_cs[1] + kz * q_wz[1]), q_sc.y);
          v.z = mul_rn((float)(a[2] - q_wz[2] * q_rs - q_azp * q_cs[2] + kz * q_wz[2]), q_sc.z);
          v.w = mul_rn((float)(a[3] - q_wz[3] * q_rs - q_azp * q_cs[3] + kz * q_wz[3]), q_sc.w);
        } else {
          v = *reinterpret_cast<const float4a*>(rq);
        }
        asm volatile("" ::: "memory");
        v.x = (v.x + b4.x) * sc; v.y = (v.y + b4.y) * sc; v.z = (v.z + b4.z) * sc; v.w = (v.w + b4.w) * sc;
        v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
        if constexpr (I8) {
          if (m < p.M) {
            trk_lo = fminf(fminf(trk_lo, v.x), fminf(fminf(v.y, v.z), v.w));
            trk_hi = fmaxf(fmaxf(trk_hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
          }
        }
        if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc32 + n) = v;
        if (p.out_f16) {
          const h4 hv = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
          *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = hv;
          if (p.f16_lo_off > 0)        // math_mode 3: the next product's operand pair (as in the direct epilogue)
            *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + n + p.f16_lo_off) =
                h4{(half_t)((v.x - (float)hv[0]) * 2048.f), (half_t)((v.y - (float)hv[1]) * 2048.f),
                   (half_t)((v.z - (float)hv[2]) * 2048.f), (half_t)((v.w - (float)hv[3]) * 2048.f)};
        }
        if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most 4 chunks of loads in flight (VGPRs)
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
  };

  // tile end: pack the finished tile (fast path) or run the direct epilogue, then re-arm the
  // accumulators for the next tile and prefetch the bias of the tile after it
  const bool fast0 = fast_kind;
  auto wave_fast = [&](int tile) __attribute__((always_inline)) -> bool {
    const int tn = tile - (tile / p.tiles_n) * p.tiles_n;
    return fast0 && tile < total_tiles && tn * BN + wn * 64 + 64 <= p.N;
  };
  int ep_tile = slot;
  auto tile_end = [&]() __attribute__((always_inline)) {
    flush();                                          // only non-empty when nk < 8
    const int tile = ep_tile;
    ep_tile += G;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM + wm * WM, n0 = tn * BN + wn * 64;
    if (wave_fast(tile)) {
      const float sc = (n0 < p.scale_cols) ? p.scale : 1.f;     // scale_cols is a multiple of 64
      if constexpr (I8) {
        // dequantise out of the three column lines (this tile's, fetched a tile ago): the same operations in the same
        // order as the fp32-result epilogues — float(exact integer) * (a_scale * w_scale[n]), + bias, * scale, ReLU
        int rs[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) rs[i] = p.q_rowsum[m0 + i * 32 + (lane & 31)];     // rows up to the padded M are readable
        const char* ln = scr + 1152 + 16 * lh;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const i4x dz = *reinterpret_cast<const i4xa*>(ln + (j * 32 + 8 * g) * 4);
            const float4 ws = *reinterpret_cast<const float4a*>(ln + 256 + (j * 32 + 8 * g) * 4);
            const float4 bb = *reinterpret_cast<const float4a*>(ln + 512 + (j * 32 + 8 * g) * 4);
            const float wsv[4] = {ws.x, ws.y, ws.z, ws.w}, bbv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int wz = (int)(int8_t)(dz[e] & 255), d = dz[e] >> 8;
                const int iv = acc[i][j][4 * g + e] - wz * rs[i] - q_azp * d;
                float x = mul_rn((float)iv, mul_rn(q_ascale, wsv[e]));
                x = mul_rn(add_rn(x, bbv[e]), sc);
                v[e] = fmaxf(x, lo);
              }
              if (m0 + i * 32 + (lane & 31) < p.M) {            // pad rows hold whatever the operand buffer held
                trk_lo = fminf(fminf(trk_lo, v[0]), fminf(fminf(v[1], v[2]), v[3]));
                trk_hi = fmaxf(fmaxf(trk_hi, v[0]), fmaxf(fmaxf(v[1], v[2]), v[3]));
              }
              hq[i][j][g] = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            }
          }
        asm volatile("" ::: "memory");
      } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            hq[i][j][g] = h4{(half_t)fmaxf(acc[i][j][4 * g + 0] * sc, lo), (half_t)fmaxf(acc[i][j][4 * g + 1] * sc, lo),
                             (half_t)fmaxf(acc[i][j][4 * g + 2] * sc, lo), (half_t)fmaxf(acc[i][j][4 * g + 3] * sc, lo)};
      }
      if constexpr (BLK) {
        // D^T fragment -> blocked layout: lanes 0..31 (rows) x {lh} (column half) = 512 contiguous bytes per store,
        // no transposition; the 16*MI stores are spread over the next k-steps, two per step (blk_store)
        ob = reinterpret_cast<char*>(p.out_f16) + ((size_t)(m0 >> 5) * (size_t)(p.N >> 3) * 32 + (lane & 31)) * 16 + lh * 8 +
             (size_t)(n0 >> 3) * 512;
        pend = 4 * MI;
      } else {
        op = p.out_f16 + (size_t)(m0 + (lane >> 3)) * p.ldc16 + n0 + (lane & 7) * 8;
        pend = 4 * MI;
      }
    } else if (KIND == 2 && m0 + WM <= p.M && n0 + 64 <= p.N) {
      lds_epilogue32(tile);
    } else {
      direct_epilogue(tile);
    }
    init_acc(wave_fast(ep_tile));                     // bias line holds the NEXT tile's bias (fetched a tile ago)
    // the line's reads (here, or the int8 column lines above) have EXECUTED before the DMA that refills it is issued —
    // waited for, not timed (the same class of hazard as the fragment reads, see the group B loop)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (I8) { if (fast0 && ep_tile < total_tiles) fetch_bias(ep_tile); }       // int8: the lines hold the CURRENT tile's constants
    else { if (fast0 && ep_tile + G < total_tiles) fetch_bias(ep_tile + G); }
  };

  // K-loop wrap: the partial sum so far (the x3 cross terms, carried 2^11 too large) is brought to the scale of the terms that follow
  auto wrap_scale_acc = [&]() __attribute__((always_inline)) {
    if constexpr (KIND == 2 && !I8) {
      const float ws = p.wrap_scale;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] *= ws;
    }
  };

  // 16 MFMAs with the step's DMA pieces slotted between them; `mid` (counted wait + phase barrier)
  // runs after 12 MFMAs so the matrix pipe does not drain at the phase boundary
  auto burst = [&](int pass_idx, auto&& mid) __attribute__((always_inline)) {      // pass_idx < 0: no pass this step
    __builtin_amdgcn_s_setprio(1);
    int piece = 0;
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (s * MI + i == (MI == 2 ? 1 : 0) && pass_idx >= 0 && !BLK) {
          __builtin_amdgcn_sched_barrier(0);
          pass_write(pass_idx);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (s * MI + i == (MI == 2 ? 3 : 1) && pass_idx >= 0 && !BLK) {
          __builtin_amdgcn_sched_barrier(0);
          pass_read(pass_idx);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (s * MI + i == KSUB * MI - 2) {
          issue_advance();
          // every fragment read of this step has executed before the phase barrier in `mid`: behind that barrier the
          // OTHER group issues DMA pieces into the slot these fragments came from (the compiler's own counted lgkmcnt waits
          // only cover the fragments of the MFMAs issued so far)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          mid();
        }
        if constexpr (!(ABL & 4))
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (I8)
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i4x, bf[s][j]), __builtin_bit_cast(i4x, af[s][i]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
        }
        // all LPS pieces go out in the KSUB*MI - 2 iterations before `mid` (which advances the cursor)
        constexpr int PPI = (LPS + KSUB * MI - 3) / (KSUB * MI - 2);
#pragma unroll
        for (int pp = 0; pp < PPI; ++pp)
          if (piece < LPS) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(piece++);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    __builtin_amdgcn_s_setprio(0);
  };

  int k = 0;                                          // flat k-step index
  int s_prev = 0, s_prev2 = 0;                        // deferred stores issued in the last two steps
  if (fast0) { fetch_bias(slot); wait_vmcnt<0>(); asm volatile("" ::: "memory"); }
  init_acc(wave_fast(slot));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!I8) { if (fast0 && slot + G < total_tiles) fetch_bias(slot + G); }

  if (grp == 0) {
    // ================= group A =================
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_step();
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < n_my; ++t) {
      for (int kt = 0; kt < nk; ++kt, ++k) {
        if (KIND == 2 && !I8 && kt == p.k_wrap && kt) wrap_scale_acc();
        const int sA = pend > 0 ? 1 : 0;
        const int pidx = sA ? 4 * MI - pend : -1;
        pend -= sA;
        load_frags();
        __builtin_amdgcn_s_barrier();
        burst(pidx, [&]() __attribute__((always_inline)) {
          if (sA) { if constexpr (BLK) blk_store(pidx); else pass_store(); }
          // younger than DMA(k+1) [issued in COMPUTE(k-1)]: store(k-1), DMA(k+2), store(k)
          const int ns = (sA + s_prev) * (BLK ? 2 : 1);              // blocked: two stores per step
          if (ns == 4) wait_vmcnt<WAITN + 4>();
          else if (ns == 2) wait_vmcnt<WAITN + 2>();
          else if (ns == 1) wait_vmcnt<WAITN + 1>();
          else wait_vmcnt<WAITN>();
          s_prev = sA;
          __builtin_amdgcn_s_barrier();
        });
      }
      tile_end();
    }
  } else {
    // ================= group B (one phase behind A) =================
#pragma unroll
    for (int s = 0; s < S; ++s) issue_step();
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < n_my; ++t) {
      for (int kt = 0; kt < nk; ++kt, ++k) {
        if (KIND == 2 && !I8 && kt == p.k_wrap && kt) wrap_scale_acc();
        const int sB = pend > 0 ? 1 : 0;
        const int pidx = sB ? 4 * MI - pend : -1;
        pend -= sB;
        load_frags();
        {
          // younger than DMA(k+1) [issued in COMPUTE(k-2)]: store(k-2), DMA(k+2), store(k-1)
          const int ns = (s_prev + s_prev2) * (BLK ? 2 : 1);
          if (ns == 4) wait_vmcnt<WAITN + 4>();
          else if (ns == 2) wait_vmcnt<WAITN + 2>();
          else if (ns == 1) wait_vmcnt<WAITN + 1>();
          else wait_vmcnt<WAITN>();
          s_prev2 = s_prev; s_prev = sB;
        }
        // This group's DMA of COMPUTE(k) refills the very slot whose fragments were requested above (stage k + 3 -> slot
        // k % 3): every wave's fragment reads must have EXECUTED before any wave of the group may issue a piece, i.e. before
        // the barrier.  Without this wait the order held only by timing (a ds_read retires long before a DMA round trip) —
        // until another kernel's workgroup shared the CU and kept the LDS pipe busy: wrong products (tools/interfere.py).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        burst(pidx, [&]() __attribute__((always_inline)) {
          if (sB) { if constexpr (BLK) blk_store(pidx); else pass_store(); }
          if (k + 1 < T) __builtin_amdgcn_s_barrier();
        });
      }
      tile_end();
    }
  }
  flush();                                            // the last tile's passes
  wait_vmcnt<0>();                                    // clamped tail DMA must land before LDS is released
  if constexpr (I8) if (p.q_part) {
    // the range of this workgroup's results, as the f16 values the consumer will read (rounding is monotone); workgroups
    // that own no tile returned above: the launcher zeroes the pairs beyond the grid
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      trk_lo = fminf(trk_lo, __shfl_xor(trk_lo, o, 64));
      trk_hi = fmaxf(trk_hi, __shfl_xor(trk_hi, o, 64));
    }
    __builtin_amdgcn_s_barrier();                     // every wave is past its last use of the ring
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) { red[2 * wave] = trk_lo; red[2 * wave + 1] = trk_hi; }
    __syncthreads();
    if (tid == 0) {
      float l = 0.f, h = 0.f;
      for (int w = 0; w < NW; ++w) { l = fminf(l, red[2 * w]); h = fmaxf(h, red[2 * w + 1]); }
      p.q_part[2 * bid] = (float)(half_t)l;
      p.q_part[2 * bid + 1] = (float)(half_t)h;
    }
    if (bid == 0)                                     // the consumer folds 256 pairs whatever this grid was
      for (int i = G + tid; i < 256; i += 512) { p.q_part[2 * i] = 0.f; p.q_part[2 * i + 1] = 0.f; }
  }
}

template <int KIND, int MI>
__global__ __launch_bounds__(512, 1) void gemm_f16_pp3(GemmDev p) { gemm_pp3_impl<KIND, MI, false>(p); }
template <int MI>
__global__ __launch_bounds__(512, 1) void gemm_i8_pp3(GemmDev p) { gemm_pp3_impl<2, MI, true>(p); }
template <int MI>
__global__ __launch_bounds__(512, 1) void gemm_i8f_pp3(GemmDev p) { gemm_pp3_impl<1, MI, true>(p); }   // f16-only results, deferred packed epilogue

static thread_local const char* g_last_gemm_kernel = "";
const char* last_gemm_kernel() { return g_last_gemm_kernel; }
void note_gemm_kernel(const char* name) { g_last_gemm_kernel = name; }

void launch_gemm(hipStream_t s, const GemmArgs& a) {
  PF_CHECK(a.K % 64 == 0 && a.K > 0, PF_ERR_INVALID_ARG, "gemm: K must be a multiple of 64");
  PF_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, PF_ERR_INVALID_ARG, "gemm: lda/ldw must be multiples of 8");
  PF_CHECK((!a.out_f32 || a.ldc32 % 4 == 0) && (!a.out_f16 || a.ldc16 % 4 == 0) && (!a.resid || a.ldr % 4 == 0) &&
               (!a.add2 || a.ld2 % 4 == 0) && a.scale_cols % 64 == 0,
           PF_ERR_INVALID_ARG, "gemm: output leading dimensions must keep 16-byte row alignment");
  GemmDev d{};
  d.A = a.A; d.W = a.W; d.bias = a.bias;
  d.out_f32 = a.out_f32; d.out_f16 = a.out_f16; d.resid = a.resid; d.add2 = a.add2;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc32 = a.ldc32; d.ldc16 = a.ldc16; d.ldr = a.ldr; d.ld2 = a.ld2;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.out_padded = a.out_padded;
  d.out_blocked = a.out_blocked; d.a_blocked = a.a_blocked;
  d.f16_lo_off = a.f16_lo_off;
  PF_CHECK(a.f16_lo_off == 0 || (a.out_f16 && !a.out_blocked && a.f16_lo_off % 4 == 0 && a.M > gemm_small_max_rows()),
           PF_ERR_INVALID_ARG, "gemm: the x3 pair output is an option of the fp32-kind epilogue (M above the short-input threshold)");
  d.k_wrap = a.k_wrap; d.a_wrap = a.a_wrap * 2; d.w_wrap = a.w_wrap * 2; d.wrap_scale = a.wrap_scale;
  PF_CHECK(a.k_wrap == 0 || (a.k_wrap > 0 && a.k_wrap < a.K / 64 && a.a_wrap % 64 == 0 && a.w_wrap % 64 == 0 && a.a_wrap <= a.k_wrap * 64 &&
                             a.w_wrap <= a.k_wrap * 64 && !a.a_blocked && !a.out_blocked && a.M > gemm_small_max_rows()),
           PF_ERR_INVALID_ARG, "gemm: K-loop wrap needs 0 < k_wrap < K / 64, cursor steps of whole k-steps inside the range walked so far, row-major operands, M above the short-input threshold");
  PF_CHECK(!a.out_blocked || (a.out_f16 && !a.out_f32 && !a.resid && !a.add2 && a.out_padded && a.N % 64 == 0),
           PF_ERR_INVALID_ARG, "gemm: blocked output needs an f16-only padded result with N % 64 == 0");
  PF_CHECK(!a.a_blocked || a.K % 64 == 0, PF_ERR_INVALID_ARG, "gemm: blocked A operand needs K % 64 == 0");
  if (a.force_mi == 4 || (a.force_mi == 0 && a.small_ws && a.M <= gemm_small_max_rows() && !a.out_blocked && !a.a_blocked)) {
    // short inputs: one-shot bricks over the whole device instead of a handful of tiles walking K (k_gemm_small.hip)
    GemmSmallArgs g{};
    g.A = a.A; g.lda = a.lda; g.W = a.W; g.ldw = a.ldw; g.bias = a.bias; g.M = a.M; g.N = a.N; g.K = a.K;
    g.out_f32 = a.out_f32; g.ldc32 = a.ldc32; g.out_f16 = a.out_f16; g.ldc16 = a.ldc16;
    g.resid = a.resid; g.ldr = a.ldr; g.add2 = a.add2; g.ld2 = a.ld2;
    g.relu = a.relu; g.scale_cols = a.scale_cols; g.scale = a.scale; g.ws = a.small_ws;
    const bool can = !a.out_blocked && !a.a_blocked && gemm_small_applicable(g);
    PF_CHECK(a.force_mi != 4 || can, PF_ERR_INVALID_ARG, "gemm: the short-input kernel does not apply to this problem");
    if (can) { launch_gemm_small(s, g); return; }
  }
  // 128-row tiles when 256-row tiles would leave CUs idle (decoder GEMMs with N = 512: 84 tiles); measured
  // A/B in one session: x = 0.9 -> 14.85 ms/step, x = 0 -> 15.15, x >= 1.5 (also the encoder N = 512 GEMMs) -> 15.9.
  // Round 3: x = 0.6 — between 0.6 and 0.9 of the CUs in 256-row tiles, the 128-row form needs a second round (SenseVoice's
  // FFN-down at M = 10 880: 172 tiles -> 340 = 2 rounds x 0.58): 3.6 -> 2.87 ms per step there, 13.55 -> 12.8 ms for configs[2]
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  dev &= 63;
  // per-device launch facts, initialised once per device under a lock: engines on different devices launch
  // from different host threads (one engine per GPU, include/paraformer_hip.h pf_group_*)
  static std::mutex init_mu;
  static int cus[64] = {0};
  // PF_GEMM_MI_X: tuning knob for tools/ (tiles < x * CUs -> 128-row tiles)
  static const float mi_x = [] { const char* e = getenv("PF_GEMM_MI_X"); return e ? (float)atof(e) : 0.6f; }();
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!cus[dev]) {
      hipDeviceProp_t prop;
      PF_HIP(hipGetDeviceProperties(&prop, dev));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_pp3<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_pp3<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_pp3<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_pp3<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_pp3<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_pp3<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1)));
      cus[dev] = cu_limit(prop.multiProcessorCount);
    }
  }
  {
    // blocked-layout results (FFN-up): the persistent 256 x 256-tile kernel (k_gemm_big.hip); PF_BIGP=0 keeps this file's kernel
    static const int use_bigp = [] { const char* e = getenv("PF_BIGP"); return (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }();   // 2: whenever it applies and fills the chip once
    const bool can = gemm_bigp_applicable(a);
    PF_CHECK(a.force_mi != 5 || can, PF_ERR_INVALID_ARG, "gemm: the persistent 256 x 256 kernel does not apply to this problem");
    // by rounds: a 256 x 256 tile costs ~1.9 tiles of this file's kernel; whichever schedule has less idle tail wins
    // (M = 16000: 504 tiles = 2 rounds vs 1008 = 4 -> persistent; SenseVoice M = 10944: 344 = 2 rounds vs 688 = 3 -> this kernel)
    const int t_big = cdiv(a.M, 256) * (a.N / 256), t_pp3 = cdiv(a.M, 256) * cdiv(a.N, GEMM_BN);
    const bool fewer_rounds = t_big >= cus[dev] && 1.9 * cdiv(t_big, cus[dev]) <= (double)cdiv(t_pp3, cus[dev]);
    if (can && (a.force_mi == 5 || (a.force_mi == 0 && use_bigp && (fewer_rounds || (use_bigp == 2 && t_big >= cus[dev]))))) {
      launch_gemm_bigp(s, a, cus[dev]);
      return;
    }
  }
  // tile height by rounds: a 128-row tile costs ~0.58 of a 256-row one (half the MFMAs, two thirds of the operand
  // bytes); whichever schedule has the shorter last round wins (decoder FFN-up, M = 5344: 336 tiles = 2 rounds vs
  // 672 = 3 x 0.58).  PF_GEMM_ROUNDS=0 keeps the tile-count rule alone.
  static const int by_rounds = [] { const char* e = getenv("PF_GEMM_ROUNDS"); return (e && e[0] == '0') ? 0 : 1; }();
  const int t2 = cdiv(d.M, 256) * cdiv(d.N, GEMM_BN), t1 = cdiv(d.M, 128) * cdiv(d.N, GEMM_BN);
  const bool few = (float)t2 < mi_x * cus[dev];
  const bool rounds1 = by_rounds && 0.58 * cdiv(t1, cus[dev]) < (double)cdiv(t2, cus[dev]);
  const int mi = a.force_mi ? (a.force_mi == 1 ? 1 : 2) : ((few || rounds1) ? 1 : 2);
  PF_CHECK(a.force_mi != 6, PF_ERR_UNSUPPORTED, "gemm: the k-step-32 kernel was removed in round 5 (numbers: profiles/round4_k32_microbench.txt)");
  d.tiles_m = cdiv(d.M, 128 * mi);
  d.tiles_n = cdiv(d.N, GEMM_BN);
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  int grid = cus[dev];
  if (grid > total) grid = total;
  const bool f16_only = a.out_f16 && !a.out_f32 && !a.resid && !a.add2 && a.out_padded && ((a.ldc16 & 7) == 0 || a.out_blocked) &&
                        a.f16_lo_off == 0 && a.k_wrap == 0;     // (pair output and the K-loop wrap live in the fp32-kind kernel)
  const int lds = gemm_lds_bytes(mi);
  note_gemm_kernel(f16_only && a.out_blocked ? (mi == 2 ? "gemm_f16_pp3<3, 2>" : "gemm_f16_pp3<3, 1>")
                   : f16_only ? (mi == 2 ? "gemm_f16_pp3<1, 2>" : "gemm_f16_pp3<1, 1>")
                              : (mi == 2 ? "gemm_f16_pp3<2, 2>" : "gemm_f16_pp3<2, 1>"));
  if (f16_only && a.out_blocked) {
    if (mi == 2) hipLaunchKernelGGL((gemm_f16_pp3<3, 2>), dim3(grid), dim3(512), lds, s, d);
    else hipLaunchKernelGGL((gemm_f16_pp3<3, 1>), dim3(grid), dim3(512), lds, s, d);
  } else if (f16_only) {
    if (mi == 2) hipLaunchKernelGGL((gemm_f16_pp3<1, 2>), dim3(grid), dim3(512), lds, s, d);
    else hipLaunchKernelGGL((gemm_f16_pp3<1, 1>), dim3(grid), dim3(512), lds, s, d);
  } else {
    if (mi == 2) hipLaunchKernelGGL((gemm_f16_pp3<2, 2>), dim3(grid), dim3(512), lds, s, d);
    else hipLaunchKernelGGL((gemm_f16_pp3<2, 1>), dim3(grid), dim3(512), lds, s, d);
  }
  PF_HIP(hipGetLastError());
}

void launch_gemm_i8(hipStream_t s, const GemmI8Args& a) {
  PF_CHECK(a.Kpad % 128 == 0 && a.Kpad > 0 && a.K > 0 && a.K <= a.Kpad, PF_ERR_INVALID_ARG, "gemm_i8: Kpad must be a multiple of 128 covering K");
  PF_CHECK(a.lda % 16 == 0 && a.ldw % 16 == 0 && a.lda >= a.Kpad && a.ldw >= a.Kpad, PF_ERR_INVALID_ARG, "gemm_i8: lda / ldw must be multiples of 16 >= Kpad");
  PF_CHECK((!a.out_f32 || a.ldc32 % 4 == 0) && (!a.out_f16 || a.ldc16 % 4 == 0) && (!a.resid || a.ldr % 4 == 0) &&
               (!a.add2 || a.ld2 % 4 == 0) && a.scale_cols % 64 == 0,
           PF_ERR_INVALID_ARG, "gemm_i8: output leading dimensions must keep 16-byte row alignment");
  PF_CHECK(a.rowsum && a.colsum && a.wzp && a.wscale && a.aparams && (a.out_f32 || a.out_f16), PF_ERR_INVALID_ARG, "gemm_i8: missing operand");
  GemmDev d{};
  d.A = reinterpret_cast<const half_t*>(a.A); d.W = reinterpret_cast<const half_t*>(a.W); d.bias = a.bias;
  d.out_f32 = a.out_f32; d.out_f16 = a.out_f16; d.resid = a.resid; d.add2 = a.add2;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc32 = a.ldc32; d.ldc16 = a.ldc16; d.ldr = a.ldr; d.ld2 = a.ld2;
  d.M = a.M; d.N = a.N; d.K = a.Kpad;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.out_padded = 0; d.out_blocked = 0; d.a_blocked = 0;
  d.q_rowsum = a.rowsum; d.q_colsum = a.colsum; d.q_wzp = a.wzp; d.q_wscale = a.wscale; d.q_aparams = a.aparams; d.q_k = a.K;
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  dev &= 63;
  static std::mutex init_mu;
  static int cus[64] = {0};
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!cus[dev]) {
      hipDeviceProp_t prop;
      PF_HIP(hipGetDeviceProperties(&prop, dev));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_i8_pp3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_i8_pp3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_i8f_pp3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2)));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_i8f_pp3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1)));
      cus[dev] = cu_limit(prop.multiProcessorCount);
    }
  }
  const int t2 = cdiv(d.M, 256) * cdiv(d.N, GEMM_BN), t1 = cdiv(d.M, 128) * cdiv(d.N, GEMM_BN);
  const bool few = (float)t2 < 0.9f * cus[dev];       // (int8: 0.9 measured better than the f16 kernels' 0.6 — SenseVoice 16.9 vs 17.5 ms)
  const bool rounds1 = 0.58 * cdiv(t1, cus[dev]) < (double)cdiv(t2, cus[dev]);
  const int mi = (few || rounds1) ? 1 : 2;
  d.tiles_m = cdiv(d.M, 128 * mi);
  d.tiles_n = cdiv(d.N, GEMM_BN);
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  const int grid = std::min(cus[dev], total);
  PF_CHECK(!a.range_out || grid <= 256, PF_ERR_UNSUPPORTED, "gemm_i8: range output needs a grid of at most 256 workgroups");
  d.q_dz = a.dz; d.q_part = a.range_out;
  // f16-only results into a padded buffer: the deferred packed epilogue (gemm_i8f_pp3), as the f16 path's KIND 1
  const bool f16_only = a.out_f16 && !a.out_f32 && !a.resid && !a.add2 && a.out_padded && (a.ldc16 & 7) == 0 && a.dz;
  d.out_padded = a.out_padded;
  if (f16_only) {
    note_gemm_kernel(mi == 2 ? "gemm_i8f_pp3<2>" : "gemm_i8f_pp3<1>");
    if (mi == 2) hipLaunchKernelGGL((gemm_i8f_pp3<2>), dim3(grid), dim3(512), gemm_lds_bytes(2), s, d);
    else hipLaunchKernelGGL((gemm_i8f_pp3<1>), dim3(grid), dim3(512), gemm_lds_bytes(1), s, d);
  } else {
    note_gemm_kernel(mi == 2 ? "gemm_i8_pp3<2>" : "gemm_i8_pp3<1>");
    if (mi == 2) hipLaunchKernelGGL((gemm_i8_pp3<2>), dim3(grid), dim3(512), gemm_lds_bytes(2), s, d);
    else hipLaunchKernelGGL((gemm_i8_pp3<1>), dim3(grid), dim3(512), gemm_lds_bytes(1), s, d);
  }
  PF_HIP(hipGetLastError());
}

}  // namespace pf
