// k_ffn.hip — the tail of a SAN-M encoder layer in ONE launch per 64-row tile (round 5).  Core: the WHOLE position-wise
// feed-forward block; template options put the attention out-projection (+ bias + residual + FSMN memory + LayerNorm norm2) in
// FRONT of it (OP) and the next layer's fused Q | K | V projection BEHIND it (QK), so that an encoder layer is attention + this
// launch and neither LayerNorm result, nor the 2048-wide hidden, nor x_mid ever visits HBM.
//
//   h = relu(xn W1^T + b1)                     xn [M,512] f16 (LayerNorm norm2 of the residual stream), W1 [2048,512]
//   x = x + h W2^T + b2                        W2 [512,2048]; x fp32 residual stream
//   n = LayerNorm_next(x) * gamma + beta       -> f16 operand of the next layer's QKV projection (and / or fp32)
//
// i.e. the MatMul + Add + Relu + MatMul + Add + Add + LayerNormalization nodes of one SAN-M encoder layer of the graph
// executed behind AliParaformerAsr/OfflineProjOfParaformer.cs:68.  Before: FFN-up (gemm_bigp_kernel, 40.6 us) wrote the
// [16000 x 2048] f16 hidden to HBM (65.5 MB) and the row-complete FFN-down (52.2 us) read it back — 131 MB of the ~600 MB an
// encoder layer moved, and two launches.  Here the hidden never leaves the compute unit.
//
// Geometry: one workgroup = 64 rows (M = 16 000 -> 250 workgroups on 256 CUs, one round), 8 wavefronts, hidden in 8 chunks
// of 256 columns.  Per chunk c
//   U: wave w computes H^T[32 hidden (its eighth of the chunk) x 64 rows] = W1[c,w] * xn^T over K = 512: 32 k-steps of
//      2 MFMA 32x32x16, the bias is added at the end of the phase; xn comes from the resident 64 KB LDS tile, the W1 fragment of a
//      k-step is used by this wave ONLY;
//      relu -> f16 -> one 16-byte LDS cell per (row, 8 hidden): the cell IS the B-operand fragment of the second product
//      (the k index of an MFMA is a summation index: W2's columns are permuted once at load so that its fragments meet
//      the order in which a D^T accumulator holds the hidden — no transposition anywhere);
//   D: wave w accumulates Y^T[64 output columns (its eighth) x 64 rows] += W2[w, c] * H^T: 16 k-steps of 4 MFMA; again
//      the W2 fragments are private to the wave.
// Because a 64-row tile leaves every weight fragment to exactly one wave, staging W through LDS would be a private FIFO
// that costs LDS capacity (the xn tile and the hidden need 128 KB) and LDS bandwidth for nothing: W1 and W2 are PRE-TILED
// at load into fragment order (1 KiB per wave-instruction, 32 KiB sequential per (chunk, wave)) and read straight into
// registers with global_load_dwordx4, PF issue positions ahead of their use.  tools/ubench/ldbw3.hip: an L2-resident
// stream reaches 114 GB/s per CU through that path with 6 loads per wave in flight (126 through LDS-DMA); a tile needs
// 4 MiB in ~40 us = 100 GB/s.  The chunk loop is FULLY unrolled: in straight-line code the compiler's s_waitcnt insertion
// counts the loads in flight exactly (across a loop back edge it falls back to vmcnt(0), which would drain the stream
// at every chunk).
// Epilogue = k_gemm_rc.hip's: the 64 x 512 fp32 tile goes through LDS, every wave owns 8 complete rows: bias + residual
// -> x, LayerNorm (two-pass statistics on DPP wave sums) -> f16 / fp32.
#include "kernels.h"

#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float4 __attribute__((may_alias)) float4a;

struct FfnDev {
  const half_t* A; const half_t* W1t; const half_t* W2t; const float* b1; const float* b2;
  const float* resid; float* out_x;
  const float* ln_g; const float* ln_b; half_t* out_n16; float* out_n32;
  int lda, ldr, ldx, ldn16, ldn32;
  int M, rot_mask;
  float eps;
  // OP = 1 (attention out-projection in front of the block, one launch for two thirds of an encoder layer):
  //   x_mid = resid + ctx Wo^T + bo + FSMN(V);  A = LayerNorm_2(x_mid) never leaves LDS, x_mid goes back into the accumulators;  x = x_mid + FFN(A)
  const half_t* ctx; const half_t* Wot; const float* bo;      // ctx [M,512] f16 (row stride lda_c), Wo in fragment order, bias
  const half_t* fsmn_v; const float* fsmn_wT;                 // V slice [M, ldv] f16, taps [11][512]
  const float* ln2_g; const float* ln2_b;                     // norm2
  int lda_c, ldv, T;
  // QK = 1 (behind the block, same launch): the NEXT layer's fused Q | K | V projection of n = LayerNorm_next(x), which then
  // never visits HBM either: Q (scaled) and K leave in the blocked [Mpad, 1024] layout, V row-major [Mpad, ldvo] (k_gemm_qkv.hip's)
  const half_t* Wqt; const float* bq; half_t* out_qk; half_t* out_v;
  int ldvo; float qscale;
  // SP > 0 (decoder form): the hidden range is split over S workgroups per tile, SP chunks each; a workgroup leaves
  // its partial product rows (fp32, no bias) and the row sums / sums of squares of ITS part of the hidden
  float* part; float* stats;                                   // [S][Mp][512], [S][Mp][2]
  int S, Mp;
};

// -DFF_TIMING (tools/ffn_timing.sh; timing experiments only): every wave of the encoder form (OP = 1) stamps the shader clock at the
// phase boundaries below (ticks since its own start) and leaves them in ff_tm[workgroup][wave][stamp]
#ifdef FF_TIMING
__device__ unsigned ff_tm[256 * 8 * 16];
#ifdef FF_TIMING_MAIN      // the 16 stamps = end of phase U / end of phase D of the 8 chunks of the main loop instead
#define FF_TS(i)
#define FF_TSM(i) { if (OP == 1 && SP == 0) tm_[i] = (unsigned)(__builtin_readcyclecounter() - t_begin_); }
#else
#define FF_TS(i) { if (OP == 1 && SP == 0) tm_[i] = (unsigned)(__builtin_readcyclecounter() - t_begin_); }
#define FF_TSM(i)
#endif
#else
#define FF_TS(i)
#define FF_TSM(i)
#endif

// fragments in flight per wave in the out-projection prologue (PFO) and in the Q | K | V tail (PFQ); the main loop's is the template's PF
#ifndef FF_PFO
#define FF_PFO 8
#endif
#ifndef FF_PFQ
#define FF_PFQ 8
#endif
constexpr int FF_BM = 64, FF_D = 512, FF_F = 2048, FF_HC = 256, FF_NC = FF_F / FF_HC;   // 8 chunks
constexpr int FF_A_BYTES = FF_BM * FF_D * 2;              // 64 KiB: 8 k-blocks of [64 rows][128 B]
constexpr int FF_H_BYTES = FF_BM * FF_HC * 2;             // 32 KiB per hidden buffer (two of them)
constexpr int FF_XROW = FF_D * 4 + 16;                    // epilogue tile row: 16-byte skew (conflict-free dump)
constexpr int FF_B_OFF = FF_BM * FF_XROW;                  // b1 (8 KiB) behind the 64 x 2064-byte fp32 tile of the epilogues (which re-use the front)
constexpr int FF_LDS = FF_B_OFF + FF_F * 4;                // 140 288 B
static_assert(FF_A_BYTES + 2 * FF_H_BYTES <= FF_B_OFF, "tile + hidden buffers must end before the bias copy");

__device__ __forceinline__ void ff_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ float ff_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}
// LDS traffic between waves: writes retired before the barrier, nothing moved across it (local address space only: the
// weight loads in flight must NOT be drained)
__device__ __forceinline__ void ff_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// FSMN memory of 8 consecutive output rows x 4 columns of one lane (k_gemm_rc.hip's rc_fsmn, 11 taps): taps reaching outside
// the utterance contribute nothing; MASKED only for rows near an utterance edge or the end of the buffer (wave-uniform)
template <bool MASKED>
__device__ __forceinline__ void ff_fsmn(float4 (&x)[8], const h4 (&win)[18], const float* __restrict__ wT, int mb, int t_first, int T, int M) {
  constexpr int FK = 11, left = 5;
  float4 w[FK];
#pragma unroll
  for (int j = 0; j < FK; ++j) w[j] = *reinterpret_cast<const float4*>(wT + (size_t)j * FF_D);
#pragma unroll
  for (int s = 0; s < 8 + FK - 1; ++s) {
    const h4 hv = win[s];
    float4 xf = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
    if (MASKED) {
      const int mm = mb - left + s;
      if (mm < 0 || mm >= M) xf = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const int r = s - j;
      if (r >= 0 && r < 8) {
        bool ok = true;
        if (MASKED) {
          int t_out = t_first + r;
          t_out = t_out >= T ? t_out - T : t_out;
          const int t_in = t_out + j - left;
          ok = t_in >= 0 && t_in < T;
        }
        if (ok) { x[r].x += w[j].x * xf.x; x[r].y += w[j].y * xf.y; x[r].z += w[j].z * xf.z; x[r].w += w[j].w * xf.w; }
      }
    }
    const int rc = s - left;
    if (rc >= 0 && rc < 8) {
      bool ok = true;
      if (MASKED) { const int mm = mb + rc; ok = mm < M; }
      if (ok) { x[rc].x += xf.x; x[rc].y += xf.y; x[rc].z += xf.z; x[rc].w += xf.w; }
    }
  }
}

// PF: how many weight fragments (1 KiB per wave each) a wave keeps in flight ahead of the one it multiplies
// ABL (tools/ffn_abl.sh; results are garbage, only the times matter): bit 0 = no weight loads in the main loop, bit 1 = no
// LDS fragment reads in the main loop, bit 2 = no MFMA, bit 3 = no chunk barriers
// XD: how many k-steps ahead of their MFMAs the LDS fragment reads are issued (ring of XD + 1 fragment pairs)
// OP: 1 = the attention out-projection (+ bias + residual + FSMN memory + LayerNorm norm2) runs in front of the block on the
// same 64 rows: its result is the block's LDS operand tile and never visits HBM as f16
// OP: 2 (split form only) = the decoder's cross-attention out-projection of the PREVIOUS layer in front of the block:
// x = resid + ctx Wo^T + bo (written by the tile's first share; out_x must not alias resid: the other shares read it),
// operand = LayerNorm norm1(x); no FSMN, nothing returns into the accumulators (the decoder block has no residual)
// SP: 0 = the encoder form (all 8 chunks, bias + residual + LayerNorm epilogue); > 0 = the decoder form: this workgroup walks SP
// chunks of the hidden starting at chunk (blockIdx % S) * SP (the weight images carry a ninth, all-zero chunk so that 3 x 3 covers
// 8), collects sum / sum of squares of its relu'd hidden rows on the way and leaves raw partial rows: the LayerNorm over the
// 2048 hidden columns that sits between the decoder's two products is applied AFTERWARDS (ffn_dec_finish_kernel) as
//   LN(h) W2^T = rstd (h (gamma (.) W2)^T - mean colsum(gamma (.) W2)) + beta W2^T
template <int PF, int ABL = 0, int XD = 2, int OP = 0, int QK = 0, int SP = 0>
__global__ __launch_bounds__(512, 1) void ffn_fused_kernel(FfnDev p) {
  static_assert((SP == 0 && OP != 2) || (SP != 0 && OP != 1 && QK == 0), "the split form: optional plain out-projection prologue (OP = 2), no tail");
  constexpr int NCH = SP ? SP : FF_NC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, l31 = lane & 31;
#ifdef FF_TIMING
  unsigned tm_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin_ = __builtin_readcyclecounter();
#endif
  const int tile = SP ? (int)blockIdx.x / p.S : (int)blockIdx.x;
  const int c_begin = SP ? ((int)blockIdx.x - tile * p.S) * SP : 0;
  const int m0 = tile * FF_BM;
  const int rot = SP ? 0 : (int)(blockIdx.x >> 3) & p.rot_mask;     // chunk order rotated per workgroup (spreads the L2 lines in time)
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 1) & 7; };

  // ---- weight stream: position gp = c * 64 + pos; pos 0..31 = W1 fragment of k-step pos, 32..63 = W2 fragment (t, j) = ((pos - 32) / 2, pos & 1)
  // (uniform base in SGPRs + the lane's 32-bit byte offset: the saddr form of global_load, no 64-bit VGPR pointers)
  const half_t* w1u = p.W1t + (size_t)wave * (32 * 512);
  const half_t* w2u = p.W2t + (size_t)wave * (32 * 512);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto wload = [&](int gp) __attribute__((always_inline)) -> h8 {
    const int c = gp >> 6, pos = gp & 63;
    const int cc = SP ? c_begin + c : (c + rot) & (FF_NC - 1);
    const half_t* b = (pos < 32 ? w1u : w2u) + (size_t)cc * (8 * 32 * 512) + (pos & 31) * 512;
    return *reinterpret_cast<const h8*>(reinterpret_cast<const char*>(b) + lane16);
  };
  constexpr int NPOS = NCH * 64;
  h8 ring[PF];
  if (!OP) {
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wload(i);
  }

  // ---- the 64 x 512 xn tile (OP: the attention context tile) -> LDS (8 k-blocks of [64 rows][128 B], 16-byte chunks XOR-swizzled by row: the fragment
  //      reads below are conflict-free); wave w brings rows 8w .. 8w+7 of every k-block
  {
    const int srow = lane >> 3, schunk = lane & 7;
    const int row = wave * 8 + srow;
    const char* src = OP ? reinterpret_cast<const char*>(p.ctx + (size_t)(m0 + row) * p.lda_c) + ((schunk ^ swz(row)) << 4)
                         : reinterpret_cast<const char*>(p.A + (size_t)(m0 + row) * p.lda) + ((schunk ^ swz(row)) << 4);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) ff_glds16(src + kb * 128, smem + kb * 8192 + wave * 1024);
    // b1 (2048 floats) as it is: read back per chunk with ds_read — a global load used right behind its issue would make
    // the compiler wait for vmcnt(0), i.e. drain the weight stream
    if (!OP) ff_glds16(reinterpret_cast<const char*>(p.b1) + wave * 1024 + lane * 16, smem + FF_B_OFF + wave * 1024);
    if (SP && !OP && wave == 0) ff_glds16(reinterpret_cast<const char*>(p.b1) + 8192 + lane * 16, smem + FF_B_OFF + 8192);   // the zero chunk's bias
  }
  // fragment read offsets of the xn tile: row half i, 16-byte k-group (2 ss + lh) of a k-block
  unsigned xo[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int ra = i * 32 + l31;
      xo[i][ss] = (unsigned)(ra * 128 + (((2 * ss + lh) ^ swz(ra)) << 4));
    }
  const unsigned ho = (unsigned)(FF_A_BYTES + lane * 16);          // hidden cells: ((i * 16 + t) * 2 + lh) * 512 + row * 16
  f16x yacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) yacc[i][j][e] = 0.f;

  if constexpr (OP != 0) {
    // ---- P1: Y^T[64 out columns of this wave x 64 rows] = Wo[wave] ctx^T over K = 512: 32 k-steps of 4 MFMAs, Wo fragments
    //      (2 per step, private to the wave) straight from their fragment-ordered image, ctx fragments from the LDS tile
    const half_t* wou = p.Wot + (size_t)wave * (64 * 512);
    // (four different K-loop starting points for four groups of workgroups — the tail's remedy against lockstep cold misses — buy
    //  0.8 k of this phase's 15 k cycles: not kept, profiles/round6_ffn_timeline.md)
    auto woload = [&](int pos) __attribute__((always_inline)) -> h8 {
      return *reinterpret_cast<const h8*>(reinterpret_cast<const char*>(wou + pos * 512) + lane16);
    };
    auto xblk = [&](int s_) __attribute__((always_inline)) -> int { return (s_ >> 2) * 8192; };
    constexpr int PFO = FF_PFO;
    h8 oring[PFO];
#pragma unroll
    for (int i = 0; i < PFO; ++i) oring[i] = woload(i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the context tile (and the first PF fragments) have landed
    __builtin_amdgcn_s_barrier();
    FF_TS(0)
    {
      constexpr int XR = XD + 1;
      h8 xf[XR][2];
#pragma unroll
      for (int s0 = 0; s0 < XD; ++s0)
#pragma unroll
        for (int i = 0; i < 2; ++i) xf[s0][i] = *reinterpret_cast<const h8*>(smem + xo[i][s0 & 3] + xblk(s0));
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        if (s + XD < 32) {
#pragma unroll
          for (int i = 0; i < 2; ++i) xf[(s + XD) % XR][i] = *reinterpret_cast<const h8*>(smem + xo[i][(s + XD) & 3] + xblk(s + XD));
        }
        h8 wo[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wo[j] = oring[(2 * s + j) % PFO];
          if (2 * s + j + PFO < 64) oring[(2 * s + j) % PFO] = woload(2 * s + j + PFO);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) yacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wo[j], xf[s % XR][i], yacc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    FF_TS(1)
    // ---- P2 (k_gemm_rc.hip's epilogue): the fp32 tile through LDS; wave w owns rows 8w .. 8w+7 completely
    const int r0 = wave * 8, mb = m0 + r0;
    float4 xv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        xv[h][r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.resid && mb + r < p.M) xv[h][r] = *reinterpret_cast<const float4*>(p.resid + (size_t)(mb + r) * p.ldr + h * 256 + 4 * lane);
      }
    h4 vwin[2][OP == 1 ? 18 : 1];
    const int t_first = OP == 1 ? mb % p.T : 0;
    const bool interior = OP == 1 && t_first >= 5 && t_first + 7 + 5 < p.T && mb + 7 + 5 < p.M;
    if constexpr (OP == 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 18; ++s) {
          int mm = mb - 5 + s;
          mm = mm < 0 ? 0 : (mm >= p.M ? p.M - 1 : mm);
          vwin[h][s] = *reinterpret_cast<const h4*>(p.fsmn_v + (size_t)mm * p.ldv + h * 256 + 4 * lane);
        }
    }
    ff_lds_barrier();                                              // every wave has finished reading the context tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* rowp = smem + (size_t)(i * 32 + l31) * FF_XROW + (wave * 64 + 4 * lh) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<float4a*>(rowp + (j * 32 + 8 * g) * 4) =
              make_float4(yacc[i][j][4 * g + 0], yacc[i][j][4 * g + 1], yacc[i][j][4 * g + 2], yacc[i][j][4 * g + 3]);
          yacc[i][j][4 * g + 0] = 0.f; yacc[i][j][4 * g + 1] = 0.f; yacc[i][j][4 * g + 2] = 0.f; yacc[i][j][4 * g + 3] = 0.f;
        }
    }
    ff_lds_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = h * 256 + 4 * lane;
      const float4 b4 = *reinterpret_cast<const float4*>(p.bo + col);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float4 v = *reinterpret_cast<const float4a*>(smem + (size_t)(r0 + r) * FF_XROW + col * 4);
        xv[h][r].x += v.x + b4.x; xv[h][r].y += v.y + b4.y; xv[h][r].z += v.z + b4.z; xv[h][r].w += v.w + b4.w;
      }
      if constexpr (OP == 1) {
        if (interior) ff_fsmn<false>(xv[h], vwin[h], p.fsmn_wT + col, mb, t_first, p.T, p.M);
        else ff_fsmn<true>(xv[h], vwin[h], p.fsmn_wT + col, mb, t_first, p.T, p.M);
      } else {
        if (c_begin == 0) {                                        // the tile's first share writes the residual stream
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (mb + r < p.M) *reinterpret_cast<float4*>(p.out_x + (size_t)(mb + r) * p.ldx + col) = xv[h][r];
        }
      }
    }
#ifdef FF_TIMING
    asm volatile("" :: "v"(xv[0][0].x), "v"(xv[1][7].w));
#endif
    FF_TS(2)
    // x_mid is the residual of the block's end: instead of a round trip through HBM it goes back into the ACCUMULATORS — the
    // second product then accumulates on top of it (rows -> LDS fp32 tile -> D^T fragments, the dump above in reverse)
    if constexpr (OP == 1) {
    ff_lds_barrier();                                              // every wave has its rows of the fp32 tile in registers
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        *reinterpret_cast<float4a*>(smem + (size_t)(r0 + r) * FF_XROW + (h * 256 + 4 * lane) * 4) = xv[h][r];
    ff_lds_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* rowp = smem + (size_t)(i * 32 + l31) * FF_XROW + (wave * 64 + 4 * lh) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t = *reinterpret_cast<const float4a*>(rowp + (j * 32 + 8 * g) * 4);
          yacc[i][j][4 * g + 0] = t.x; yacc[i][j][4 * g + 1] = t.y; yacc[i][j][4 * g + 2] = t.z; yacc[i][j][4 * g + 3] = t.w;
        }
    }
    }
    // LayerNorm norm2 of the complete rows -> f16 -> the block's operand tile (swizzled k-block layout)
    float4 g4[2], be4[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      g4[h] = *reinterpret_cast<const float4*>(p.ln2_g + h * 256 + 4 * lane);
      be4[h] = *reinterpret_cast<const float4*>(p.ln2_b + h * 256 + 4 * lane);
    }
    float mean[8], rstd[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float sm = ((xv[0][r].x + xv[0][r].y) + (xv[0][r].z + xv[0][r].w)) + ((xv[1][r].x + xv[1][r].y) + (xv[1][r].z + xv[1][r].w));
      mean[r] = ff_wave_sum(sm) * (1.0f / FF_D);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float m = mean[r];
      xv[0][r].x -= m; xv[0][r].y -= m; xv[0][r].z -= m; xv[0][r].w -= m;
      xv[1][r].x -= m; xv[1][r].y -= m; xv[1][r].z -= m; xv[1][r].w -= m;
      const float q = ((xv[0][r].x * xv[0][r].x + xv[0][r].y * xv[0][r].y) + (xv[0][r].z * xv[0][r].z + xv[0][r].w * xv[0][r].w)) +
                      ((xv[1][r].x * xv[1][r].x + xv[1][r].y * xv[1][r].y) + (xv[1][r].z * xv[1][r].z + xv[1][r].w * xv[1][r].w));
      rstd[r] = 1.0f / sqrtf(ff_wave_sum(q) * (1.0f / FF_D) + p.eps);
    }
    ff_lds_barrier();                                              // every wave has read its fragments of x_mid back
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = r0 + r;
      const float k = rstd[r];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 d = xv[h][r];
        const h4 y = h4{(half_t)(d.x * k * g4[h].x + be4[h].x), (half_t)(d.y * k * g4[h].y + be4[h].y),
                        (half_t)(d.z * k * g4[h].z + be4[h].z), (half_t)(d.w * k * g4[h].w + be4[h].w)};
        const int c = h * 256 + 4 * lane;                          // column -> k-block c >> 6, 16-byte chunk (c & 63) >> 3, 8 bytes inside it
        *reinterpret_cast<h4*>(smem + (c >> 6) * 8192 + row * 128 + ((((c & 63) >> 3) ^ swz(row)) << 4) + (c & 7) * 2) = y;
      }
    }
    // b1 -> LDS now (its place was inside the fp32 tile's footprint? no: behind it — but the DMA is cheapest here, off the
    // critical path), then the block's first weight fragments
    ff_glds16(reinterpret_cast<const char*>(p.b1) + wave * 1024 + lane * 16, smem + FF_B_OFF + wave * 1024);
    if (SP && wave == 0) ff_glds16(reinterpret_cast<const char*>(p.b1) + 8192 + lane * 16, smem + FF_B_OFF + 8192);
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wload(i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ff_lds_barrier();                                              // the operand tile is complete
    FF_TS(3)
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the tile (and the first PF weight fragments) have landed
    __builtin_amdgcn_s_barrier();
  }

  float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};          // SP: sum / sum of squares of the hidden, rows i * 32 + l31, this lane's share
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int cc = SP ? c_begin + c : (c + rot) & (FF_NC - 1);
    // ---- U: hidden^T[32 x 64] of this wave, accumulators start as the bias
    f16x hacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) hacc[i][e] = 0.f;
    constexpr int XR = XD + 1;
    h8 xf[XR][2];
#pragma unroll
    for (int s0 = 0; s0 < XD; ++s0)
#pragma unroll
      for (int i = 0; i < 2; ++i) xf[s0][i] = *reinterpret_cast<const h8*>(smem + xo[i][s0 & 3] + (s0 >> 2) * 8192);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int gp = c * 64 + s;
      if (s + XD < 32 && !(ABL & 2)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xf[(s + XD) % XR][i] = *reinterpret_cast<const h8*>(smem + xo[i][(s + XD) & 3] + ((s + XD) >> 2) * 8192);
      } else if (s + XD < 32) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xf[(s + XD) % XR][i] = xf[s % XR][i];
      }
      const h8 w = ring[gp % PF];
      if (!(ABL & 1) && gp + PF < NPOS) ring[gp % PF] = wload(gp + PF);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!(ABL & 4)) hacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, xf[s % XR][i], hacc[i], 0, 0, 0);
        else { hacc[i][0] += (float)w[0]; hacc[i][1] += (float)xf[s % XR][i][0]; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    FF_TSM(2 * c)
    // relu -> f16 -> the cells of k-steps t = 2 wave, 2 wave + 1 of the second product
    {
      char* hb = smem + ho + (c & 1) * FF_H_BYTES;
      float4 bias4[4];                                     // hidden columns 8 g + 4 lh + 0..3 of this wave's 32
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bias4[g] = *reinterpret_cast<const float4a*>(smem + FF_B_OFF + (cc * FF_HC + wave * 32 + 8 * g + 4 * lh) * 4);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          h8 v;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 b4 = bias4[2 * tt + (q >> 2)];
            const float f = hacc[i][8 * tt + q] + ((q & 3) == 0 ? b4.x : (q & 3) == 1 ? b4.y : (q & 3) == 2 ? b4.z : b4.w);
            v[q] = (half_t)(f > 0.f ? f : 0.f);
          }
          *reinterpret_cast<h8*>(hb + (i * 16 + 2 * wave + tt) * 1024) = v;
        }
    }
    if (!(ABL & 8)) ff_lds_barrier();
    // ---- D: Y^T[64 x 64] of this wave += W2[wave, chunk] * hidden^T
    {
      const char* hb = smem + ho + (c & 1) * FF_H_BYTES;
      h8 hf[XR][2];
#pragma unroll
      for (int t0 = 0; t0 < XD; ++t0)
#pragma unroll
        for (int i = 0; i < 2; ++i) hf[t0][i] = *reinterpret_cast<const h8*>(hb + (i * 16 + t0) * 1024);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int gp = c * 64 + 32 + 2 * t;
        if (t + XD < 16 && !(ABL & 2)) {
#pragma unroll
          for (int i = 0; i < 2; ++i) hf[(t + XD) % XR][i] = *reinterpret_cast<const h8*>(hb + (i * 16 + t + XD) * 1024);
        } else if (t + XD < 16) {
#pragma unroll
          for (int i = 0; i < 2; ++i) hf[(t + XD) % XR][i] = hf[t % XR][i];
        }
        h8 w2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          w2[j] = ring[(gp + j) % PF];
          if (!(ABL & 1) && gp + j + PF < NPOS) ring[(gp + j) % PF] = wload(gp + j + PF);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (!(ABL & 4)) yacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[j], hf[t % XR][i], yacc[i][j], 0, 0, 0);
            else { yacc[i][j][0] += (float)w2[j][0]; yacc[i][j][1] += (float)hf[t % XR][i][0]; }
          }
        if (SP && (t & 7) == wave) {
          // every wave reads every cell of the hidden: wave w keeps the statistics of k-steps w and w + 8 (exact f16 products, fp32 sums)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const h2 pr = {hf[t % XR][i][2 * q], hf[t % XR][i][2 * q + 1]};
              st_s[i] = __builtin_amdgcn_fdot2(pr, h2{(_Float16)1.f, (_Float16)1.f}, st_s[i], false);
              st_q[i] = __builtin_amdgcn_fdot2(pr, pr, st_q[i], false);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    FF_TSM(2 * c + 1)
  }

  FF_TS(4)
  // ---- epilogue (k_gemm_rc.hip's): wave w owns rows 8w .. 8w+7 completely; lane: columns 4 lane and 256 + 4 lane
  // (the lane index is re-derived here with mbcnt: keeping the work-item id alive across the unrolled main loop costs
  //  two spilled registers, i.e. a scratch arena, for nothing)
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int lh_e = lane_e >> 5, l31_e = lane_e & 31;
  const int r0 = wave * 8;
  const int mb = m0 + r0;
  float4 xv[2][8];
  float4 xs_[QK ? 2 : 1][QK ? 8 : 1];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      xv[h][r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (OP) {                                            // x_mid is already inside the accumulators
      } else if (p.resid && mb + r < p.M) {
        xv[h][r] = *reinterpret_cast<const float4*>(p.resid + (size_t)(mb + r) * p.ldr + h * 256 + 4 * lane_e);
      }
    }
  ff_lds_barrier();                                        // every wave has finished reading the hidden and the xn tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    char* rowp = smem + (size_t)(i * 32 + l31_e) * FF_XROW + (wave * 64 + 4 * lh_e) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4a*>(rowp + (j * 32 + 8 * g) * 4) =
            make_float4(yacc[i][j][4 * g + 0], yacc[i][j][4 * g + 1], yacc[i][j][4 * g + 2], yacc[i][j][4 * g + 3]);
  }
  if constexpr (SP != 0) {
    // the 16 (wave, lane half) shares of the row statistics -> the place of b1 (dead by now): [wave][lh][64 rows] float2
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float2*>(smem + FF_B_OFF + (((wave * 2 + lh_e) * 64 + i * 32 + l31_e) << 3)) = make_float2(st_s[i], st_q[i]);
    __syncthreads();
    const int split = c_begin / SP;
    if (wave == 0 && m0 + lane_e < p.M) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float2 v = *reinterpret_cast<const float2*>(smem + FF_B_OFF + ((k * 64 + lane_e) << 3));
        s += v.x; q += v.y;
      }
      *reinterpret_cast<float2*>(p.stats + ((size_t)split * p.Mp + m0 + lane_e) * 2) = make_float2(s, q);
    }
    float* prow = p.part + ((size_t)split * p.Mp + mb) * FF_D + 4 * lane_e;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (mb + r >= p.M) break;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<float4*>(prow + (size_t)r * FF_D + h * 256) =
            *reinterpret_cast<const float4a*>(smem + (size_t)(r0 + r) * FF_XROW + (h * 256 + 4 * lane_e) * 4);
    }
    return;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int col = h * 256 + 4 * lane_e;
    const float4 b4 = *reinterpret_cast<const float4*>(p.b2 + col);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = *reinterpret_cast<const float4a*>(smem + (size_t)(r0 + r) * FF_XROW + col * 4);
      xv[h][r].x += v.x + b4.x; xv[h][r].y += v.y + b4.y; xv[h][r].z += v.z + b4.z; xv[h][r].w += v.w + b4.w;
    }
    if (QK != 0 && p.Wqt) {
      // with a tail behind it the residual stream is stored at the very END of the kernel: vmcnt retires stores in order with the
      // loads, so 128 KB of x stores per workgroup issued here would sit in front of every counted wait of the tail's weight stream
      // (and of the vmcnt(0) that precedes it) until HBM has acknowledged them — 64 registers held instead (round 6: −4 %)
#pragma unroll
      for (int r = 0; r < 8; ++r) xs_[h][r] = xv[h][r];
    } else if (p.out_x) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (mb + r < p.M) *reinterpret_cast<float4*>(p.out_x + (size_t)(mb + r) * p.ldx + col) = xv[h][r];
    }
  }
  FF_TS(5)
  if (!p.ln_g) return;
  float4 g4[2], be4[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    g4[h] = *reinterpret_cast<const float4*>(p.ln_g + h * 256 + 4 * lane_e);
    be4[h] = *reinterpret_cast<const float4*>(p.ln_b + h * 256 + 4 * lane_e);
  }
  float mean[8], rstd[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float s = ((xv[0][r].x + xv[0][r].y) + (xv[0][r].z + xv[0][r].w)) + ((xv[1][r].x + xv[1][r].y) + (xv[1][r].z + xv[1][r].w));
    mean[r] = ff_wave_sum(s) * (1.0f / FF_D);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float m = mean[r];
    xv[0][r].x -= m; xv[0][r].y -= m; xv[0][r].z -= m; xv[0][r].w -= m;
    xv[1][r].x -= m; xv[1][r].y -= m; xv[1][r].z -= m; xv[1][r].w -= m;
    const float q = ((xv[0][r].x * xv[0][r].x + xv[0][r].y * xv[0][r].y) + (xv[0][r].z * xv[0][r].z + xv[0][r].w * xv[0][r].w)) +
                    ((xv[1][r].x * xv[1][r].x + xv[1][r].y * xv[1][r].y) + (xv[1][r].z * xv[1][r].z + xv[1][r].w * xv[1][r].w));
    rstd[r] = 1.0f / sqrtf(ff_wave_sum(q) * (1.0f / FF_D) + p.eps);
  }
  h4 yh[QK ? 2 : 1][QK ? 8 : 1];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = mb + r;
    if (QK || m < p.M) {
      const float k = rstd[r];
      const float4 d0 = xv[0][r], d1 = xv[1][r];
      const float4 y0 = make_float4(d0.x * k * g4[0].x + be4[0].x, d0.y * k * g4[0].y + be4[0].y,
                                    d0.z * k * g4[0].z + be4[0].z, d0.w * k * g4[0].w + be4[0].w);
      const float4 y1 = make_float4(d1.x * k * g4[1].x + be4[1].x, d1.y * k * g4[1].y + be4[1].y,
                                    d1.z * k * g4[1].z + be4[1].z, d1.w * k * g4[1].w + be4[1].w);
      if (QK) {
        yh[0][r] = h4{(half_t)y0.x, (half_t)y0.y, (half_t)y0.z, (half_t)y0.w};
        yh[1][r] = h4{(half_t)y1.x, (half_t)y1.y, (half_t)y1.z, (half_t)y1.w};
      }
      if (m >= p.M) continue;
      if (p.out_n16) {
        half_t* o = p.out_n16 + (size_t)m * p.ldn16 + 4 * lane_e;
        *reinterpret_cast<h4*>(o) = h4{(half_t)y0.x, (half_t)y0.y, (half_t)y0.z, (half_t)y0.w};
        *reinterpret_cast<h4*>(o + 256) = h4{(half_t)y1.x, (half_t)y1.y, (half_t)y1.z, (half_t)y1.w};
      }
      if (p.out_n32) {
        float* o = p.out_n32 + (size_t)m * p.ldn32 + 4 * lane_e;
        *reinterpret_cast<float4*>(o) = y0;
        *reinterpret_cast<float4*>(o + 256) = y1;
      }
    }
  }
  if constexpr (QK != 0) {
    if (!p.Wqt) return;
    // ---- the next layer's Q | K | V projection of these 64 rows: n -> LDS operand tile, three passes of 64 output columns per
    //      wave (Q, K, V: 3 x 8 x 64 = 1536), each the out-projection's loop over K = 512 with Wq fragments from their image
    ff_lds_barrier();                                      // every wave has read its rows of the fp32 tile
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = r0 + r, c = h * 256 + 4 * lane_e;
        *reinterpret_cast<h4*>(smem + (c >> 6) * 8192 + row * 128 + ((((c & 63) >> 3) ^ swz(row)) << 4) + (c & 7) * 2) = yh[h][r];
      }
    unsigned xq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int ra = i * 32 + l31_e;
        xq[i][ss] = (unsigned)(ra * 128 + (((2 * ss + lh_e) ^ swz(ra)) << 4));
      }
    const unsigned lane16e = (unsigned)lane_e * 16u;
    // bq (1536 floats) -> the LDS place of b1 (dead by now): a global load used right behind its issue would drain the stream
    if (wave < 6) ff_glds16(reinterpret_cast<const char*>(p.bq) + wave * 1024 + lane_e * 16, smem + FF_B_OFF + wave * 1024);
    // ONE weight stream over the three passes (192 fragments of this wave), PF ahead across the pass ends
    const half_t* wqu = p.Wqt + (size_t)wave * (64 * 512);
    // The workgroups of an XCD walk Q, K, V in three different orders (round 6).  Every layer's Wq image is COLD in the XCD's L2
    // when the tail starts; with one order all 32 CUs run into every first-touch miss together and wait for it together (three
    // passes 18 + 18 + 15 k cycles for 8 k cycles of MFMA each).  With three orders each third of the image is fetched by one group
    // while the other two stream something else, and a workgroup's second and third pass find their lines already filled by the
    // other groups: 14 + 6.6 + 5.3 k cycles, −8 % on the whole launch (profiles/round6_ffn_timeline.md).  The results do not
    // depend on the order.
    const int prot = (int)((blockIdx.x >> 3) % 3u);
    auto wqload = [&](int gp) __attribute__((always_inline)) -> h8 {      // gp = pass * 64 + pos
      int pq = (gp >> 6) + prot;
      pq = pq >= 3 ? pq - 3 : pq;
      return *reinterpret_cast<const h8*>(reinterpret_cast<const char*>(wqu + (size_t)pq * (8 * 64 * 512) + (gp & 63) * 512) + lane16e);
    };
    constexpr int PFQ = FF_PFQ;
    h8 qring[PFQ];
#pragma unroll
    for (int i = 0; i < PFQ; ++i) qring[i] = wqload(i);
    FF_TS(6)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ff_lds_barrier();                                      // the operand tile and the bias are complete
    FF_TS(7)
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      int pr = pass + prot;                               // which of Q | K | V this pass computes
      pr = pr >= 3 ? pr - 3 : pr;
      // accumulators start as the bias of this wave's 64 columns
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4a*>(smem + FF_B_OFF + (pr * 512 + wave * 64 + 4 * lh_e + j * 32 + 8 * g) * 4);
#pragma unroll
          for (int i = 0; i < 2; ++i) { yacc[i][j][4 * g + 0] = b4.x; yacc[i][j][4 * g + 1] = b4.y; yacc[i][j][4 * g + 2] = b4.z; yacc[i][j][4 * g + 3] = b4.w; }
        }
      {
        constexpr int XR = XD + 1;
        h8 xf[XR][2];
#pragma unroll
        for (int s0 = 0; s0 < XD; ++s0)
#pragma unroll
          for (int i = 0; i < 2; ++i) xf[s0][i] = *reinterpret_cast<const h8*>(smem + xq[i][s0 & 3] + (s0 >> 2) * 8192);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          if (s + XD < 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[(s + XD) % XR][i] = *reinterpret_cast<const h8*>(smem + xq[i][(s + XD) & 3] + ((s + XD) >> 2) * 8192);
          }
          h8 wq[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int gp = pass * 64 + 2 * s + j;
            wq[j] = qring[gp % PFQ];
            if (gp + PFQ < 192) qring[gp % PFQ] = wqload(gp + PFQ);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) yacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[j], xf[s % XR][i], yacc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FF_TS(8 + 2 * pass)
      if (pr < 2) {
        // Q (scaled) | K: 32 x 8 blocks of the blocked [M, 1024] matrix.  Lanes l / l + 32 hold columns 0..3 / 4..7 of row l of block g;
        // after v_permlane32_swap lane l holds the whole 16-byte row of block 2 gp, lane l + 32 that of block 2 gp + 1: 8 dwordx4 per
        // pass and lane instead of 16 dwordx2 (store tails are issue-bound, MI355X guide T21).  Plain stores: on gfx950 stores count in
        // vmcnt IN ORDER with the loads, so a write-through (sc1) store's long acknowledgement holds up the next pass's counted waits
        // for its weight fragments (round 6 timeline).
        const int qk0 = pr * 512 + wave * 64;
        char* ob = reinterpret_cast<char*>(p.out_qk) + ((size_t)(m0 >> 5) * 128 + (size_t)(qk0 >> 3)) * 512 + l31_e * 16;
        constexpr size_t rb_stride = (size_t)128 * 512;
        const float sc = pr == 0 ? p.qscale : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              unsigned x[2], y[2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                f2v xa = {yacc[i][j][8 * gp + 2 * e + 0], yacc[i][j][8 * gp + 2 * e + 1]};
                f2v ya = {yacc[i][j][8 * gp + 4 + 2 * e + 0], yacc[i][j][8 * gp + 4 + 2 * e + 1]};
                xa *= sc; ya *= sc;
                x[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(xa, h2v));
                y[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(ya, h2v));
              }
#pragma unroll
              for (int e = 0; e < 2; ++e) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
              const h4 lo_ = __builtin_bit_cast(h4, (unsigned long long)x[0] | ((unsigned long long)x[1] << 32));
              const h4 hi_ = __builtin_bit_cast(h4, (unsigned long long)y[0] | ((unsigned long long)y[1] << 32));
              const h8 hv = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
              asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ob + i * rb_stride + (size_t)(j * 4 + 2 * gp + lh_e) * 512), "v"(hv) : "memory");
            }
      } else {
        // V: row-major [M, ldvo]; lanes l / l + 32 hold columns 8g + 0..3 / 8g + 4..7 of row l: after v_permlane32_swap lane l
        // holds the 8 columns of group 2gp, lane l + 32 those of group 2gp + 1 (k_gemm_qkv.hip)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          half_t* vrow = p.out_v + (size_t)(m0 + i * 32 + l31_e) * p.ldvo + wave * 64;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              unsigned x[2], y[2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const f2v xa = {yacc[i][j][8 * gp + 2 * e + 0], yacc[i][j][8 * gp + 2 * e + 1]};
                const f2v ya = {yacc[i][j][8 * gp + 4 + 2 * e + 0], yacc[i][j][8 * gp + 4 + 2 * e + 1]};
                x[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(xa, h2v));
                y[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(ya, h2v));
              }
#pragma unroll
              for (int e = 0; e < 2; ++e) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
              const h4 lo_ = __builtin_bit_cast(h4, (unsigned long long)x[0] | ((unsigned long long)x[1] << 32));
              const h4 hi_ = __builtin_bit_cast(h4, (unsigned long long)y[0] | ((unsigned long long)y[1] << 32));
              const h8 hv = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
              asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(vrow + j * 32 + 16 * gp + 8 * lh_e), "v"(hv) : "memory");
            }
        }
      }
      FF_TS(9 + 2 * pass)
    }
    if (p.out_x) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (mb + r < p.M) *reinterpret_cast<float4*>(p.out_x + (size_t)(mb + r) * p.ldx + h * 256 + 4 * lane_e) = xs_[h][r];
    }
#ifdef FF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FF_TS(14)
    if (OP == 1 && SP == 0 && lane_e == 0 && blockIdx.x < 256)
      for (int i = 0; i < 16; ++i) ff_tm[(blockIdx.x * 8 + wave) * 16 + i] = tm_[i];
#endif
  }
}

#ifdef FF_TIMING
}  // namespace pf
extern "C" int pf_debug_ffn_timing(unsigned* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::ff_tm), (size_t)n * 4, 0, hipMemcpyDeviceToHost);
}
namespace pf {
#endif

// W1 [2048, ldw1] and W2 [512, ldw2] (f16, K-contiguous) -> the fragment-ordered images the kernel streams:
//   W1t[((c * 8 + w) * 32 + s) * 512 + l * 8 + e] = W1[c * 256 + w * 32 + (l & 31)][16 s + 8 (l >> 5) + e]
//   W2t[((c * 8 + w) * 32 + 2 t + j) * 512 + l * 8 + q] = W2[w * 64 + j * 32 + (l & 31)][c * 256 + 16 t + 8 (q >> 2) + 4 (l >> 5) + (q & 3)]
// one thread per 16-byte piece (131 072 pieces each)
__global__ void ffn_retile_kernel(const half_t* __restrict__ W1, int ldw1, const half_t* __restrict__ W2, int ldw2,
                                  half_t* __restrict__ W1t, half_t* __restrict__ W2t) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * 131072) return;
  const int which = idx >> 17, pc = idx & 131071;
  const int l = pc & 63, piece = (pc >> 6) & 31, w = (pc >> 11) & 7, c = pc >> 14;
  h8 v;
  if (which == 0) {
    const half_t* src = W1 + (size_t)(c * 256 + w * 32 + (l & 31)) * ldw1 + 16 * piece + 8 * (l >> 5);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e];
    *reinterpret_cast<h8*>(W1t + (size_t)pc * 8) = v;
  } else {
    const int t = piece >> 1, j = piece & 1;
    const half_t* src = W2 + (size_t)(w * 64 + j * 32 + (l & 31)) * ldw2 + c * 256 + 16 * t + 4 * (l >> 5);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[8 * (q >> 2) + (q & 3)];
    *reinterpret_cast<h8*>(W2t + (size_t)pc * 8) = v;
  }
}

// Wo [512, ldw] -> Wot[((w * 32 + t) * 2 + j) * 512 + l * 8 + e] = Wo[w * 64 + j * 32 + (l & 31)][16 t + 8 (l >> 5) + e]  (32 768 pieces)
__global__ void ffn_retile_out_kernel(const half_t* __restrict__ Wo, int ldw, half_t* __restrict__ Wot) {
  const int pc = blockIdx.x * blockDim.x + threadIdx.x;
  if (pc >= 32768) return;
  const int l = pc & 63, j = (pc >> 6) & 1, t = (pc >> 7) & 31, w = pc >> 12;
  const half_t* src = Wo + (size_t)(w * 64 + j * 32 + (l & 31)) * ldw + 16 * t + 8 * (l >> 5);
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[e];
  *reinterpret_cast<h8*>(Wot + (size_t)pc * 8) = v;
}


// ---- decoder form (SP > 0) -----------------------------------------------------------------------------------------
// The decoder's position-wise block is  t = LN_F(relu(xn W1^T + b1)) W2^T  (LayerNorm over the 2048 hidden columns between the
// products, W2 without bias: the w_1 / norm / w_2 nodes of the decoder graph behind AliParaformerAsr/OfflineProjOfParaformer.cs:68).
// With h = relu(..) (f16, as the unfused path stores it), mean / rstd its row statistics:
//   t = rstd (h W2g^T - mean c) + d,   W2g = gamma (.) W2 (rounded to f16 ONCE, from the fp32 tensor),  c = colsum(W2g) (of the
//   rounded values: the subtraction then cancels exactly what the product accumulated),  d = W2 beta
// so the chunked kernel applies unchanged and the statistics are only needed at the end.  Image layout (f16 unless noted):
//   W1t [9 chunks][8 waves][32 steps][512]  |  W2gt [9][8][32][512]  |  b1p fp32 [2304]  |  c fp32 [512]  |  d fp32 [512]
// chunk 8 = zeros (3 splits x 3 chunks).
constexpr size_t FFD_W = (size_t)9 * 16384 * 8;            // halves per weight image
size_t ffn_dec_image_bytes() { return 2 * FFD_W * 2 + (2304 + 512 + 512) * 4; }

__global__ void ffn_dec_retile_kernel(const half_t* __restrict__ W1, int ldw1, const float* __restrict__ W2, const float* __restrict__ gamma,
                                      const float* __restrict__ b1, half_t* __restrict__ img) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int NP = 9 * 16384;
  if (idx < 2 * NP) {
    const int which = idx >= NP, pc = which ? idx - NP : idx;
    const int l = pc & 63, piece = (pc >> 6) & 31, w = (pc >> 11) & 7, c = pc >> 14;
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
    if (c < 8) {
      if (which == 0) {
        const half_t* src = W1 + (size_t)(c * 256 + w * 32 + (l & 31)) * ldw1 + 16 * piece + 8 * (l >> 5);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e];
      } else {
        const int t = piece >> 1, j = piece & 1;
        const int k0 = c * 256 + 16 * t + 4 * (l >> 5);
        const float* src = W2 + (size_t)(w * 64 + j * 32 + (l & 31)) * FF_F + k0;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (_Float16)(src[8 * (q >> 2) + (q & 3)] * gamma[k0 + 8 * (q >> 2) + (q & 3)]);
      }
    }
    *reinterpret_cast<h8*>(img + (which ? FFD_W : 0) + (size_t)pc * 8) = v;
  } else if (idx < 2 * NP + 2304) {
    const int k = idx - 2 * NP;
    reinterpret_cast<float*>(img + 2 * FFD_W)[k] = k < FF_F ? b1[k] : 0.f;
  }
}
// c[n] = sum_k float(f16(gamma_k W2[n][k])), d[n] = sum_k beta_k W2[n][k]; one 256-thread workgroup per n, sums in double
__global__ void ffn_dec_colsum_kernel(const float* __restrict__ W2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      half_t* __restrict__ img) {
  __shared__ double sc[256], sd[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  double c = 0.0, d = 0.0;
  for (int k = tid; k < FF_F; k += 256) {
    const float w = W2[(size_t)n * FF_F + k];
    c += (double)(float)(_Float16)(w * gamma[k]);
    d += (double)w * (double)beta[k];
  }
  sc[tid] = c; sd[tid] = d;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { sc[tid] += sc[tid + o]; sd[tid] += sd[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    float* f = reinterpret_cast<float*>(img + 2 * FFD_W) + 2304;
    f[n] = (float)sc[0];
    f[512 + n] = (float)sd[0];
  }
}

void launch_ffn_dec_retile(hipStream_t s, const half_t* W1, int ldw1, const float* W2_f32, const float* gamma, const float* beta,
                           const float* b1, half_t* img) {
  constexpr int n = 2 * 9 * 16384 + 2304;
  hipLaunchKernelGGL(ffn_dec_retile_kernel, dim3((n + 255) / 256), dim3(256), 0, s, W1, ldw1, W2_f32, gamma, b1, img);
  PF_HIP(hipGetLastError());
  hipLaunchKernelGGL(ffn_dec_colsum_kernel, dim3(FF_D), dim3(256), 0, s, W2_f32, gamma, beta, img);
  PF_HIP(hipGetLastError());
}

// Finishing pass of the decoder form: one wave per row.  t = rstd (sum_s part_s - mean c) + d with the row statistics of the
// whole hidden (sum over the S shares; variance in double: E[h^2] - mean^2 of 2048 non-negative values), then the LayerNorm
// behind the block (norm2 / after_norm): two-pass statistics on DPP wave sums -> fp32 and / or f16.
__global__ __launch_bounds__(256) void ffn_dec_finish_kernel(const float* __restrict__ part, const float* __restrict__ stats, int S, int Mp, int M,
                                                             const float* __restrict__ cd, float eps_f, const float* __restrict__ g,
                                                             const float* __restrict__ b, float eps, float* __restrict__ t32, int ldt,
                                                             float* __restrict__ n32, int ldn32, half_t* __restrict__ n16, int ldn16) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  float s = 0.f, q = 0.f;
  for (int sp = 0; sp < S; ++sp) {
    const float* r = part + ((size_t)sp * Mp + m) * FF_D + 4 * lane;
    const float4 v0 = *reinterpret_cast<const float4*>(r), v1 = *reinterpret_cast<const float4*>(r + 256);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    const float2 st = *reinterpret_cast<const float2*>(stats + ((size_t)sp * Mp + m) * 2);
    s += st.x; q += st.y;
  }
  const double mu_d = (double)s * (1.0 / FF_F);
  double var = (double)q * (1.0 / FF_F) - mu_d * mu_d;
  var = var > 0.0 ? var : 0.0;
  const float mu = (float)mu_d, rs = (float)(1.0 / sqrt(var + (double)eps_f));
  const float4 c0 = *reinterpret_cast<const float4*>(cd + 4 * lane), c1 = *reinterpret_cast<const float4*>(cd + 256 + 4 * lane);
  const float4 d0 = *reinterpret_cast<const float4*>(cd + 512 + 4 * lane), d1 = *reinterpret_cast<const float4*>(cd + 768 + 4 * lane);
  float4 y0 = make_float4(rs * (a0.x - mu * c0.x) + d0.x, rs * (a0.y - mu * c0.y) + d0.y, rs * (a0.z - mu * c0.z) + d0.z, rs * (a0.w - mu * c0.w) + d0.w);
  float4 y1 = make_float4(rs * (a1.x - mu * c1.x) + d1.x, rs * (a1.y - mu * c1.y) + d1.y, rs * (a1.z - mu * c1.z) + d1.z, rs * (a1.w - mu * c1.w) + d1.w);
  if (t32) {
    *reinterpret_cast<float4*>(t32 + (size_t)m * ldt + 4 * lane) = y0;
    *reinterpret_cast<float4*>(t32 + (size_t)m * ldt + 256 + 4 * lane) = y1;
  }
  if (!g) return;
  const float mean = ff_wave_sum(((y0.x + y0.y) + (y0.z + y0.w)) + ((y1.x + y1.y) + (y1.z + y1.w))) * (1.0f / FF_D);
  y0.x -= mean; y0.y -= mean; y0.z -= mean; y0.w -= mean; y1.x -= mean; y1.y -= mean; y1.z -= mean; y1.w -= mean;
  const float k = 1.0f / sqrtf(ff_wave_sum(((y0.x * y0.x + y0.y * y0.y) + (y0.z * y0.z + y0.w * y0.w)) +
                                            ((y1.x * y1.x + y1.y * y1.y) + (y1.z * y1.z + y1.w * y1.w))) * (1.0f / FF_D) + eps);
  const float4 g0 = *reinterpret_cast<const float4*>(g + 4 * lane), g1 = *reinterpret_cast<const float4*>(g + 256 + 4 * lane);
  const float4 b0 = *reinterpret_cast<const float4*>(b + 4 * lane), b1 = *reinterpret_cast<const float4*>(b + 256 + 4 * lane);
  const float4 z0 = make_float4(y0.x * k * g0.x + b0.x, y0.y * k * g0.y + b0.y, y0.z * k * g0.z + b0.z, y0.w * k * g0.w + b0.w);
  const float4 z1 = make_float4(y1.x * k * g1.x + b1.x, y1.y * k * g1.y + b1.y, y1.z * k * g1.z + b1.z, y1.w * k * g1.w + b1.w);
  if (n32) {
    *reinterpret_cast<float4*>(n32 + (size_t)m * ldn32 + 4 * lane) = z0;
    *reinterpret_cast<float4*>(n32 + (size_t)m * ldn32 + 256 + 4 * lane) = z1;
  }
  if (n16) {
    *reinterpret_cast<h4*>(n16 + (size_t)m * ldn16 + 4 * lane) = h4{(half_t)z0.x, (half_t)z0.y, (half_t)z0.z, (half_t)z0.w};
    *reinterpret_cast<h4*>(n16 + (size_t)m * ldn16 + 256 + 4 * lane) = h4{(half_t)z1.x, (half_t)z1.y, (half_t)z1.z, (half_t)z1.w};
  }
}

// how many workgroups share a tile's hidden range: the form with the shortest critical path on 256 CUs (a workgroup = one CU:
// 140 KB of LDS).  Measured at M = 5344 (profiles/round5_dec_ffn.txt): 17 us of fixed work + 4.3 us per chunk per round, a
// second round of a few workgroups costs about half a round; the finishing pass ~7 us + 1.2 us per share it sums.
int ffn_dec_splits(int M) {
  static const int forced = env_int("PF_DEC_FFN_SPLITS", 0);
  if (forced == 1 || forced == 2 || forced == 3 || forced == 4 || forced == 8) return forced;
  const int tiles = cdiv(M, FF_BM);
  const int cand[5] = {1, 2, 3, 4, 8}, nch[5] = {8, 4, 3, 2, 1};
  int best = 1; double bt = 1e30;
  for (int i = 0; i < 5; ++i) {
    const int wgs = tiles * cand[i], rounds = cdiv(wgs, 256);
    const double last = (wgs - (rounds - 1) * 256) / 256.0;            // how full the last round is
    const double t = (17.0 + 4.3 * nch[i]) * (rounds - 1 + (rounds > 1 ? 0.5 + 0.5 * last : 1.0)) + 1.2 * cand[i];
    if (t < bt - 0.5) { bt = t; best = cand[i]; }
  }
  return best;
}
size_t ffn_dec_workspace_bytes(int M, int splits) {        // partial rows + statistics of every split
  const size_t Mp = (size_t)round_up(M, 64);
  return (size_t)(splits > 0 ? splits : ffn_dec_splits(M)) * Mp * (FF_D + 2) * 4;
}

void launch_ffn_dec(hipStream_t s, const FfnDecArgs& a) {
  PF_CHECK(a.M > 0 && a.img && a.ws, PF_ERR_INVALID_ARG, "ffn_dec: missing operand");
  PF_CHECK(a.lda % 8 == 0 && (!a.t32 || a.ldt % 4 == 0) && (!a.n32 || a.ldn32 % 4 == 0) && (!a.n16 || a.ldn16 % 4 == 0), PF_ERR_INVALID_ARG,
           "ffn_dec: leading dimensions must keep 16-byte (8-byte for f16) row alignment");
  PF_CHECK(a.no_finish || (!a.ln_g == !a.ln_b && (a.ln_g || (!a.n16 && !a.n32)) && (a.t32 || a.n16 || a.n32)), PF_ERR_INVALID_ARG, "ffn_dec: outputs");
  const int S = a.splits > 0 ? a.splits : ffn_dec_splits(a.M);
  PF_CHECK(S == 1 || S == 2 || S == 3 || S == 4 || S == 8, PF_ERR_INVALID_ARG, "ffn_dec: splits must be 1 | 2 | 3 | 4 | 8");
  const int Mp = (int)round_up(a.M, 64);
  const bool op = a.ctx != nullptr;
  PF_CHECK(op || a.A, PF_ERR_INVALID_ARG, "ffn_dec: missing operand");
  PF_CHECK(!op || (a.Wot && a.bo && a.resid && a.out_x && a.out_x != a.resid && a.ln1_g && a.ln1_b && a.lda_c % 8 == 0 && a.ldr % 4 == 0 &&
                   a.ldx % 4 == 0),
           PF_ERR_INVALID_ARG, "ffn_dec: the out-projection form needs ctx, Wo, bias, the residual, a separate x output and norm1");
  FfnDev d{};
  d.A = a.A; d.lda = a.lda; d.W1t = a.img; d.W2t = a.img + FFD_W; d.b1 = reinterpret_cast<const float*>(a.img + 2 * FFD_W);
  d.M = a.M; d.S = S; d.Mp = Mp;
  d.part = reinterpret_cast<float*>(a.ws); d.stats = d.part + (size_t)S * Mp * FF_D;
  d.ctx = a.ctx; d.lda_c = a.lda_c; d.Wot = a.Wot; d.bo = a.bo; d.resid = a.resid; d.ldr = a.ldr; d.out_x = a.out_x; d.ldx = a.ldx;
  d.ln2_g = a.ln1_g; d.ln2_b = a.ln1_b; d.eps = a.eps1;
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  constexpr int LDSB = FF_LDS + 1024;
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 0, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 0, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 0, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 0, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 2, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 2, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 2, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 2, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 2, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      attr_set[dev & 63] = true;
    }
  }
  const dim3 grid((unsigned)(cdiv(a.M, FF_BM) * S));
#ifdef PF_FFN_ABLATIONS
  if (const char* e = getenv("PF_DEC_ABL")) {              // tools/dec_ffn_abl.sh: garbage results, only the times matter (S = 3 forms)
    const int abl = atoi(e);
    auto go = [&](auto kern) {
      PF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
      hipLaunchKernelGGL(kern, grid, dim3(512), LDSB, s, d);
    };
    if (abl && S == 3) {
      switch (abl) {
        case 1: op ? go(ffn_fused_kernel<8, 1, 2, 2, 0, 3>) : go(ffn_fused_kernel<8, 1, 2, 0, 0, 3>); break;
        case 3: op ? go(ffn_fused_kernel<8, 3, 2, 2, 0, 3>) : go(ffn_fused_kernel<8, 3, 2, 0, 0, 3>); break;
        case 7: op ? go(ffn_fused_kernel<8, 7, 2, 2, 0, 3>) : go(ffn_fused_kernel<8, 7, 2, 0, 0, 3>); break;
        case 15: op ? go(ffn_fused_kernel<8, 15, 2, 2, 0, 3>) : go(ffn_fused_kernel<8, 15, 2, 0, 0, 3>); break;
        default: PF_CHECK(false, PF_ERR_INVALID_ARG, "PF_DEC_ABL: 1 | 3 | 7 | 15");
      }
      PF_HIP(hipGetLastError());
      return;                                              // no finishing pass: the split kernel alone
    }
  }
#endif
#define PF_DEC_GO(OPV, SPV)                                                                            \
  do {                                                                                                 \
    note_gemm_kernel("ffn_fused_kernel<8, 0, 2, " #OPV ", 0, " #SPV ">");                              \
    hipLaunchKernelGGL((ffn_fused_kernel<8, 0, 2, OPV, 0, SPV>), grid, dim3(512), LDSB, s, d);         \
  } while (0)
  if (op) {
    switch (S) {
      case 1: PF_DEC_GO(2, 8); break;
      case 2: PF_DEC_GO(2, 4); break;
      case 3: PF_DEC_GO(2, 3); break;
      case 4: PF_DEC_GO(2, 2); break;
      default: PF_DEC_GO(2, 1); break;
    }
  } else {
    switch (S) {
      case 1: PF_DEC_GO(0, 8); break;
      case 2: PF_DEC_GO(0, 4); break;
      case 3: PF_DEC_GO(0, 3); break;
      case 4: PF_DEC_GO(0, 2); break;
      default: PF_DEC_GO(0, 1); break;
    }
  }
#undef PF_DEC_GO
  PF_HIP(hipGetLastError());
  if (a.no_finish) return;
  hipLaunchKernelGGL(ffn_dec_finish_kernel, dim3((unsigned)cdiv(a.M, 4)), dim3(256), 0, s, d.part, d.stats, S, Mp, a.M,
                     reinterpret_cast<const float*>(a.img + 2 * FFD_W) + 2304, a.eps_hidden, a.ln_g, a.ln_b, a.eps, a.t32, a.ldt, a.n32, a.ldn32,
                     a.n16, a.ldn16);
  PF_HIP(hipGetLastError());
}

const float* ffn_dec_image_cd(const half_t* img) { return reinterpret_cast<const float*>(img + 2 * FFD_W) + 2304; }

size_t ffn_fused_weight_bytes() { return (size_t)2 * 131072 * 16; }      // W1t | W2t: 2 MiB each
size_t ffn_outproj_weight_bytes() { return (size_t)32768 * 16; }          // Wot: 512 KiB

void launch_ffn_retile_out(hipStream_t s, const half_t* Wo, int ldw, half_t* Wot) {
  hipLaunchKernelGGL(ffn_retile_out_kernel, dim3(32768 / 256), dim3(256), 0, s, Wo, ldw, Wot);
  PF_HIP(hipGetLastError());
}

void launch_ffn_retile(hipStream_t s, const half_t* W1, int ldw1, const half_t* W2, int ldw2, half_t* Wt) {
  hipLaunchKernelGGL(ffn_retile_kernel, dim3(2 * 131072 / 256), dim3(256), 0, s, W1, ldw1, W2, ldw2, Wt, Wt + (size_t)131072 * 8);
  PF_HIP(hipGetLastError());
}

bool ffn_fused_applicable(int D, int F) { return D == FF_D && F == FF_F; }

void launch_ffn_fused(hipStream_t s, const FfnFusedArgs& a) {
  PF_CHECK(a.M > 0 && a.Wt && a.b1 && a.b2, PF_ERR_INVALID_ARG, "ffn_fused: missing operand");
  PF_CHECK(a.lda % 8 == 0 && (!a.resid || a.ldr % 4 == 0) && (!a.out_x || a.ldx % 4 == 0) && (!a.out_n16 || a.ldn16 % 4 == 0) &&
               (!a.out_n32 || a.ldn32 % 4 == 0),
           PF_ERR_INVALID_ARG, "ffn_fused: leading dimensions must keep 16-byte (8-byte for f16) row alignment");
  PF_CHECK(!a.ln_g == !a.ln_b && (a.ln_g || (!a.out_n16 && !a.out_n32)), PF_ERR_INVALID_ARG, "ffn_fused: LayerNorm outputs need gamma and beta");
  PF_CHECK(a.out_x || a.out_n16 || a.out_n32, PF_ERR_INVALID_ARG, "ffn_fused: no output requested");
  FfnDev d;
  d.A = a.A; d.W1t = a.Wt; d.W2t = a.Wt + (size_t)131072 * 8; d.b1 = a.b1; d.b2 = a.b2;
  d.resid = a.resid; d.out_x = a.out_x; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.out_n16 = a.out_n16; d.out_n32 = a.out_n32;
  d.lda = a.lda; d.ldr = a.ldr; d.ldx = a.ldx; d.ldn16 = a.ldn16; d.ldn32 = a.ldn32;
  d.M = a.M; d.eps = a.eps;
  d.ctx = a.ctx; d.Wot = a.Wot; d.bo = a.bo; d.fsmn_v = a.fsmn_v; d.fsmn_wT = a.fsmn_wT; d.ln2_g = a.ln2_g; d.ln2_b = a.ln2_b;
  d.lda_c = a.lda_c; d.ldv = a.ldv; d.T = a.T > 0 ? a.T : a.M;
  d.Wqt = a.Wqt; d.bq = a.bq; d.out_qk = a.out_qk; d.out_v = a.out_v; d.ldvo = a.ldvo; d.qscale = a.qscale;
  const bool op = a.ctx != nullptr;
  PF_CHECK(!a.Wqt || (op && a.bq && a.out_qk && a.out_v && a.ldvo % 8 == 0 && a.ln_g), PF_ERR_INVALID_ARG,
           "ffn_fused: the Q | K | V tail needs the out-projection form, the next LayerNorm, bias and both outputs");
  PF_CHECK(!op || (a.Wot && a.bo && a.fsmn_v && a.fsmn_wT && a.ln2_g && a.ln2_b && a.lda_c % 8 == 0 && a.ldv % 4 == 0 && d.T >= 8),
           PF_ERR_INVALID_ARG, "ffn_fused: the out-projection form needs ctx, Wo, bias, the V slice, FSMN taps and norm2");
  PF_CHECK(op || a.A, PF_ERR_INVALID_ARG, "ffn_fused: missing operand");
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  static int rot_mask = 7, pf = 8, xd = 2, abl = 0;
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<12, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)ffn_fused_kernel<8, 0, 2, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      if (const char* e = getenv("PF_FFN_ROT")) rot_mask = atoi(e) & 7;      // 0 | 1 | 3 | 7: chunk-order rotation period - 1
      if (const char* e = getenv("PF_FFN_ABL")) abl = atoi(e);
      if (const char* e = getenv("PF_FFN_XD")) xd = atoi(e);                 // LDS fragment reads 1 | 2 | 3 k-steps ahead
      if (const char* e = getenv("PF_FFN_PF")) pf = atoi(e);                 // 8 | 12 fragments in flight per wave (16 spills)
      attr_set[dev & 63] = true;
    }
  }
  d.rot_mask = rot_mask;
  const dim3 grid((unsigned)cdiv(a.M, FF_BM));
#ifdef PF_FFN_ABLATIONS
  if (abl) {
    auto go = [&](auto kern) {
      PF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
      hipLaunchKernelGGL(kern, grid, dim3(512), FF_LDS, s, d);
    };
    switch (abl) {
      case 1: go(ffn_fused_kernel<8, 1, 2>); break;
      case 2: go(ffn_fused_kernel<8, 2, 2>); break;
      case 3: go(ffn_fused_kernel<8, 3, 2>); break;
      case 4: go(ffn_fused_kernel<8, 4, 2>); break;
      case 7: go(ffn_fused_kernel<8, 7, 2>); break;
      case 8: go(ffn_fused_kernel<8, 8, 2>); break;
      case 15: go(ffn_fused_kernel<8, 15, 2>); break;
      default: PF_CHECK(false, PF_ERR_INVALID_ARG, "PF_FFN_ABL: 1 | 2 | 3 | 4 | 7 | 8 | 15");
    }
    PF_HIP(hipGetLastError());
    return;
  }
#endif
  (void)abl;
  if (op && a.Wqt) {
    note_gemm_kernel("ffn_fused_kernel<8, 0, 2, 1, 1, 0>");
    hipLaunchKernelGGL((ffn_fused_kernel<8, 0, 2, 1, 1>), grid, dim3(512), FF_LDS, s, d);
    PF_HIP(hipGetLastError());
    return;
  }
  if (op) {
    note_gemm_kernel("ffn_fused_kernel<8, 0, 2, 1, 0, 0>");
    hipLaunchKernelGGL((ffn_fused_kernel<8, 0, 2, 1>), grid, dim3(512), FF_LDS, s, d);
    PF_HIP(hipGetLastError());
    return;
  }
  if (pf >= 12) {                                       // 12 fragments in flight, LDS reads one k-step ahead
    note_gemm_kernel("ffn_fused_kernel<12, 0, 1, 0, 0, 0>");
    hipLaunchKernelGGL((ffn_fused_kernel<12, 0, 1>), grid, dim3(512), FF_LDS, s, d);
  } else if (xd >= 3) {
    note_gemm_kernel("ffn_fused_kernel<8, 0, 3, 0, 0, 0>");
    hipLaunchKernelGGL((ffn_fused_kernel<8, 0, 3>), grid, dim3(512), FF_LDS, s, d);
  } else if (xd == 2) {
    note_gemm_kernel("ffn_fused_kernel<8, 0, 2, 0, 0, 0>");
    hipLaunchKernelGGL((ffn_fused_kernel<8, 0, 2>), grid, dim3(512), FF_LDS, s, d);
  } else {
    note_gemm_kernel("ffn_fused_kernel<8, 0, 1, 0, 0, 0>");
    hipLaunchKernelGGL((ffn_fused_kernel<8, 0, 1>), grid, dim3(512), FF_LDS, s, d);
  }
  PF_HIP(hipGetLastError());
}

}  // namespace pf
