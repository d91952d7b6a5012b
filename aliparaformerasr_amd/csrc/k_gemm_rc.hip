// k_gemm_rc.hip — ROW-COMPLETE GEMM for the encoder's two N = 512 projections (attention output, FFN down):
//
//   y = A[M,K] * W[512,K]^T + bias           (f16 operands, fp32 accumulate, MFMA 32x32x16)
//   x = resid + y + FSMN(V)                   (fp32 residual stream; FSMN = 11-tap depthwise conv + identity over
//                                              the f16 V slice of the QKV buffer, per utterance of T rows)
//   n = LayerNorm(x) * gamma + beta           (-> f16 operand of the next GEMM and / or fp32)
//
// i.e. what the ONNX graph executed by InferenceSession.Run (reference call site
// AliParaformerAsr/OfflineProjOfParaformer.cs:68) spreads over MatMul + Add + Conv(depthwise) + Add + Add +
// LayerNormalization nodes, in ONE launch.  Why: with 256 x 128 tiles the out-projection moved 112 MB per launch
// through its epilogue (fp32 residual in/out + the fp32 FSMN buffer) and was followed by a LayerNorm kernel that
// re-read the 32 MB it had just written; the FSMN kernel wrote and the GEMM re-read another 64 MB per layer
// (round-1 VERDICT "What's weak" #4: ~600 MB of HBM traffic per encoder layer against ~0.5 GB per STEP algorithmic).
//
// Geometry: one workgroup = 64 rows x all 512 columns (M = 16 000 rows -> 250 workgroups for 256 CUs, one round),
// 8 wavefronts x (64 rows x 64 columns = 2 x 2 MFMA blocks).  Per k-step (BK = 64) a workgroup stages 8 KB of A and
// the 64 KB W panel through LDS with global_load_lds_dwordx4 (2-stage ring, counted vmcnt, raw s_barrier): every
// CU streams the whole W (L2-resident: 0.5 MB / 2 MB), so the main loop is bound by the L2 -> LDS feed
// (72 KB per 4.2 MFLOP), not by the matrix pipe — accepted, because row-completeness removes three kernels and
// ~250 MB of HBM traffic per layer.  Not persistent: one tile per workgroup, so in the epilogue the whole LDS is
// free: the 64 x 512 fp32 tile is written row-major into LDS (full-lane 16-byte writes, 16-byte row skew, conflict
// free), then every wavefront owns 8 COMPLETE rows: lane l holds columns 4l..4l+3 and 256+4l..; residual loads,
// x stores and the f16 stores are whole 1-2 KB row segments, the LayerNorm statistics are two wave reductions
// (two-pass: mean, then sum of squared deviations), no cross-workgroup traffic of any kind.
#include "kernels.h"

#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float4 __attribute__((may_alias)) float4a;

struct RcDev {
  const half_t* A; const half_t* W; const float* bias;
  const float* resid; float* out_x;
  const half_t* fsmn_v; const float* fsmn_wT;
  const float* ln_g; const float* ln_b; half_t* out_n16; float* out_n32;
  int lda, ldw, ldr, ldx, ldv, ldn16, ldn32;
  int M, K, T, a_blocked;
  float eps;
};

constexpr int RC_BM = 64, RC_BN = 512, RC_BK = 64;
constexpr int RC_ROWB = RC_BK * 2;                       // bytes of one tile row per k-step
constexpr int RC_A_BYTES = RC_BM * RC_ROWB;              // 8 KiB
constexpr int RC_W_BYTES = RC_BN * RC_ROWB;              // 64 KiB
constexpr int RC_STAGE = RC_A_BYTES + RC_W_BYTES;        // 72 KiB
constexpr int RC_XROW = RC_BN * 4 + 16;                  // epilogue tile: 16-byte skew per row (conflict-free dump)
constexpr int RC_LDS = 2 * RC_STAGE > RC_BM * RC_XROW ? 2 * RC_STAGE : RC_BM * RC_XROW;   // 144 KiB
constexpr int RC_TOUCH = 8 * 256;                        // + a dead 256-byte line per wave: target of the A prefetch touches (PFD > 0)

__device__ __forceinline__ void rc_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void rc_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void rc_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }   // vmcnt 63, expcnt 7, lgkmcnt 0

// wave-wide sum, broadcast to every lane: four DPP steps give every lane its 16-lane row total (VALU only, no LDS
// crossbar as ds_bpermute-based shuffles use), the four row totals are read through SGPRs
__device__ __forceinline__ float rc_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

// FSMN accumulation for 8 consecutive output rows x 4 columns of one lane.  MASKED: rows near an utterance edge
// (or the end of the buffer) — taps reaching outside the utterance contribute nothing (zero padding of the
// depthwise conv); the condition is wave-uniform, so the masks are scalar selects.
template <int FK, bool MASKED>
__device__ __forceinline__ void rc_fsmn(float4 (&x)[8], const h4 (&win)[FK > 0 ? 8 + FK - 1 : 1], const float* __restrict__ wT,
                                        int mb, int t_first, int T, int M) {
  constexpr int left = (FK - 1) / 2;
  float4 w[FK];
#pragma unroll
  for (int j = 0; j < FK; ++j) w[j] = *reinterpret_cast<const float4*>(wT + (size_t)j * RC_BN);
#pragma unroll
  for (int s = 0; s < 8 + FK - 1; ++s) {
    const h4 hv = win[s];
    float4 xf = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
    if (MASKED) {
      const int mm = mb - left + s;                        // input row: outside the buffer -> contributes nothing
      if (mm < 0 || mm >= M) xf = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const int r = s - j;                                 // output row fed by (input s, tap j)
      if (r >= 0 && r < 8) {
        bool ok = true;
        if (MASKED) {
          int t_out = t_first + r;
          t_out = t_out >= T ? t_out - T : t_out;          // 8 rows cross at most one utterance boundary (T >= 8)
          const int t_in = t_out + j - left;
          ok = t_in >= 0 && t_in < T;
        }
        if (ok) {
          x[r].x += w[j].x * xf.x; x[r].y += w[j].y * xf.y; x[r].z += w[j].z * xf.z; x[r].w += w[j].w * xf.w;
        }
      }
    }
    const int rc = s - left;                               // identity term: the row itself
    if (rc >= 0 && rc < 8) {
      bool ok = true;
      if (MASKED) { const int mm = mb + rc; ok = mm < M; }
      if (ok) { x[rc].x += xf.x; x[rc].y += xf.y; x[rc].z += xf.z; x[rc].w += xf.w; }
    }
  }
}

// FK: FSMN taps (0 = no FSMN term), compile time so the tap loop and the register window unroll.
// PFD > 0: every k-step each wave also TOUCHES an eighth of the A stage PFD k-steps ahead (one 4-byte LDS-DMA per lane into
// a dead LDS line: 64 lanes on 8 distinct 128-byte lines, the TA coalesces them) so that the stage's own DMA, two steps
// before it is consumed, finds the activation rows in L2: A is the only COLD operand of this kernel (each row block is read
// by exactly one workgroup, straight from HBM / the Infinity Cache), and with one stage in flight a cold line's latency is
// exposed every step.  The touch is a tenth vector-memory operation per wave and step: the counted waits say 10, not 9.
template <int FK, int PFD>
__global__ __launch_bounds__(512, 1) void gemm_rc_kernel(RcDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5;
  const int m0 = blockIdx.x * RC_BM;
  const int nk = p.K / RC_BK;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 1) & 7; };

  // ---- LDS-DMA source offsets (bytes, per lane) and bases (uniform)
  const int srow = lane >> 3, schunk = lane & 7;
  unsigned a_vo, w_vo[8];
  {
    const int row = wave * 8 + srow;                       // tile row of this wave's A piece
    a_vo = p.a_blocked ? (unsigned)((wave >> 2) * (p.K >> 3) * 512 + (wave & 3) * 1024 + lane * 16)
                       : (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int wr = (wave + 8 * i) * 8 + srow;            // W row (output column) of piece i
      w_vo[i] = (unsigned)(wr * p.ldw + ((schunk ^ swz(wr)) << 3)) * 2u;
    }
  }
  const char* a_base = p.a_blocked ? reinterpret_cast<const char*>(p.A) + (size_t)(m0 >> 5) * (size_t)(p.K >> 3) * 512
                                   : reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
  const char* w_base = reinterpret_cast<const char*>(p.W);
  const int a_step = p.a_blocked ? (RC_BK / 8) * 512 : RC_BK * 2;
  // touch line (wave * 8 + (lane & 7)) of the 64 lines of A stage k
  const int tl = wave * 8 + (lane & 7);
  const unsigned t_vo = p.a_blocked ? (unsigned)((tl >> 5) * (p.K >> 3) * 512 + (tl & 31) * 128) : (unsigned)(tl * p.lda * 2);
  auto touch = [&](int k) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (size_t)k * a_step + t_vo),
                                     (__attribute__((address_space(3))) void*)(smem + RC_LDS + wave * 256), 4, 0, 0);
  };
  constexpr int NOPS = PFD > 0 ? 10 : 9;                   // vector-memory operations per wave and k-step
  auto issue = [&](int k, int buf) __attribute__((always_inline)) {
    char* st = smem + buf * RC_STAGE + wave * 1024;
    rc_glds16(a_base + (size_t)k * a_step + a_vo, st);
#pragma unroll
    for (int i = 0; i < 8; ++i) rc_glds16(w_base + (size_t)k * (RC_BK * 2) + w_vo[i], st + RC_A_BYTES + i * 8192);
  };

  // ---- fragment read offsets inside a stage (bytes)
  unsigned fa[4][2], fb[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = i * 32 + (lane & 31);
    const int rb = wave * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa[s][i] = p.a_blocked ? (unsigned)((((ra >> 5) * 8 + 2 * s + lh) * 32 + (ra & 31)) * 16)
                             : (unsigned)(ra * RC_ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
      fb[s][i] = (unsigned)(RC_A_BYTES + rb * RC_ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  }

  f16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- main loop: 2-stage ring; stage k+1 stays in flight across the barriers of step k (counted vmcnt).
  // Every workgroup streams the SAME W panel: they start at different k-steps (rotation by workgroup index) so that
  // the CUs of one XCD do not hit the same L2 lines in the same microsecond; the 9 DMA pieces of a step are slotted
  // between the 16 MFMAs (issued as one burst they stall the CU's address unit in front of the matrix pipe).
  const int rot = (int)((blockIdx.x >> 3) % (unsigned)nk);
  auto kk = [&](int k) __attribute__((always_inline)) -> int { const int q = k + rot; return q >= nk ? q - nk : q; };
  issue(kk(0), 0);
  if (PFD > 0) touch(kk(PFD < nk ? PFD : nk - 1));
  if (nk > 1) issue(kk(1), 1);
  for (int k = 0; k < nk; ++k) {
    if (k + 1 < nk) rc_wait_vmcnt<NOPS>(); else rc_wait_vmcnt<0>();   // this wave's 9 pieces of stage k have landed
    __builtin_amdgcn_s_barrier();                                      // ... and everybody else's
    const char* rd = smem + (k & 1) * RC_STAGE;
    h8 af[4][2], bf[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[s][i] = *(const h8*)(rd + fa[s][i]);
        bf[s][i] = *(const h8*)(rd + fb[s][i]);
      }
    rc_wait_lgkm0();                                                   // fragments are in registers
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                      // nobody reads this stage any more
    const bool more = k + 2 < nk;
    const int kn = more ? kk(k + 2) : 0;
    char* st = smem + (k & 1) * RC_STAGE + wave * 1024;
    const char* an = a_base + (size_t)kn * a_step + a_vo;
    const char* wn_ = w_base + (size_t)kn * (RC_BK * 2);
    if (PFD > 0 && more) { const int kt = k + 1 + PFD; touch(kk(kt < nk ? kt : nk - 1)); }
    __builtin_amdgcn_s_setprio(1);
    int piece = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
        // 9 pieces over the 8 (s, i) slots: two in the first slot, one in each of the others
#pragma unroll
        for (int pp = 0; pp < (s == 0 && i == 0 ? 2 : 1); ++pp) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
            if (piece == 0) rc_glds16(an, st);
            else rc_glds16(wn_ + w_vo[piece > 0 ? piece - 1 : 0], st + RC_A_BYTES + (piece - 1) * 8192);
          }
          ++piece;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  }

  // ---- epilogue.  Wave w owns rows 8w .. 8w+7 COMPLETELY; lane: columns 4*lane and 256 + 4*lane.
  // The residual rows (and the first half of the FSMN window) are requested BEFORE the accumulators are exchanged
  // through LDS: their HBM round trip overlaps the dump and the barrier.
  const int r0 = wave * 8;
  const int mb = m0 + r0;                                  // first output row of this wave
  float4 xv[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      xv[h][r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.resid && mb + r < p.M)
        xv[h][r] = *reinterpret_cast<const float4*>(p.resid + (size_t)(mb + r) * p.ldr + h * 256 + 4 * lane);
    }
  constexpr int FKW = FK > 0 ? 8 + FK - 1 : 1;             // input rows of the FSMN window
  constexpr int left = FK > 0 ? (FK - 1) / 2 : 0, right = FK > 0 ? FK - 1 - left : 0;
  h4 vwin[2][FKW];
  int t_first = 0;
  bool interior = true;
  if constexpr (FK > 0) {
    t_first = mb % p.T;                                    // position of row mb inside its utterance
    // interior: the 8 rows and their halo lie inside ONE utterance and inside [0, M): no masks at all
    interior = t_first >= left && t_first + 7 + right < p.T && mb + 7 + right < p.M;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < FKW; ++s) {
        int mm = mb - left + s;
        mm = mm < 0 ? 0 : (mm >= p.M ? p.M - 1 : mm);      // address clamp only; validity is decided by the masks
        vwin[h][s] = *reinterpret_cast<const h4*>(p.fsmn_v + (size_t)mm * p.ldv + h * 256 + 4 * lane);
      }
  }
  // the 64 x 512 fp32 tile, row-major in LDS (every stage read has retired behind the last barrier and no DMA is
  // outstanding).  D^T fragment: lane = row (lane & 31), 4 consecutive columns per quad.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    char* rowp = smem + (size_t)(i * 32 + (lane & 31)) * RC_XROW + (wave * 64 + 4 * lh) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4a*>(rowp + (j * 32 + 8 * g) * 4) =
            make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int col = h * 256 + 4 * lane;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = *reinterpret_cast<const float4a*>(smem + (size_t)(r0 + r) * RC_XROW + col * 4);
      xv[h][r].x += v.x + b4.x; xv[h][r].y += v.y + b4.y; xv[h][r].z += v.z + b4.z; xv[h][r].w += v.w + b4.w;
    }
    if constexpr (FK > 0) {
      if (interior) rc_fsmn<FK, false>(xv[h], vwin[h], p.fsmn_wT + col, mb, t_first, p.T, p.M);
      else rc_fsmn<FK, true>(xv[h], vwin[h], p.fsmn_wT + col, mb, t_first, p.T, p.M);
    }
    if (p.out_x) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (mb + r < p.M) *reinterpret_cast<float4*>(p.out_x + (size_t)(mb + r) * p.ldx + col) = xv[h][r];
    }
  }
  if (!p.ln_g) return;

  // ---- LayerNorm of the complete rows (two-pass statistics; one wave = one row, reductions on the VALU via DPP)
  float4 g4[2], be4[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    g4[h] = *reinterpret_cast<const float4*>(p.ln_g + h * 256 + 4 * lane);
    be4[h] = *reinterpret_cast<const float4*>(p.ln_b + h * 256 + 4 * lane);
  }
  float mean[8], rstd[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float s = ((xv[0][r].x + xv[0][r].y) + (xv[0][r].z + xv[0][r].w)) + ((xv[1][r].x + xv[1][r].y) + (xv[1][r].z + xv[1][r].w));
    mean[r] = rc_wave_sum(s) * (1.0f / RC_BN);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float m = mean[r];
    xv[0][r].x -= m; xv[0][r].y -= m; xv[0][r].z -= m; xv[0][r].w -= m;
    xv[1][r].x -= m; xv[1][r].y -= m; xv[1][r].z -= m; xv[1][r].w -= m;
    const float q = ((xv[0][r].x * xv[0][r].x + xv[0][r].y * xv[0][r].y) + (xv[0][r].z * xv[0][r].z + xv[0][r].w * xv[0][r].w)) +
                    ((xv[1][r].x * xv[1][r].x + xv[1][r].y * xv[1][r].y) + (xv[1][r].z * xv[1][r].z + xv[1][r].w * xv[1][r].w));
    rstd[r] = 1.0f / sqrtf(rc_wave_sum(q) * (1.0f / RC_BN) + p.eps);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = mb + r;
    if (m < p.M) {
      const float k = rstd[r];
      const float4 d0 = xv[0][r], d1 = xv[1][r];
      const float4 y0 = make_float4(d0.x * k * g4[0].x + be4[0].x, d0.y * k * g4[0].y + be4[0].y,
                                    d0.z * k * g4[0].z + be4[0].z, d0.w * k * g4[0].w + be4[0].w);
      const float4 y1 = make_float4(d1.x * k * g4[1].x + be4[1].x, d1.y * k * g4[1].y + be4[1].y,
                                    d1.z * k * g4[1].z + be4[1].z, d1.w * k * g4[1].w + be4[1].w);
      if (p.out_n16) {
        half_t* o = p.out_n16 + (size_t)m * p.ldn16 + 4 * lane;
        *reinterpret_cast<h4*>(o) = h4{(half_t)y0.x, (half_t)y0.y, (half_t)y0.z, (half_t)y0.w};
        *reinterpret_cast<h4*>(o + 256) = h4{(half_t)y1.x, (half_t)y1.y, (half_t)y1.z, (half_t)y1.w};
      }
      if (p.out_n32) {
        float* o = p.out_n32 + (size_t)m * p.ldn32 + 4 * lane;
        *reinterpret_cast<float4*>(o) = y0;
        *reinterpret_cast<float4*>(o + 256) = y1;
      }
    }
  }
}

void launch_gemm_rc(hipStream_t s, const GemmRcArgs& a) {
  PF_CHECK(a.M > 0 && a.K > 0 && a.K % 64 == 0, PF_ERR_INVALID_ARG, "gemm_rc: K must be a positive multiple of 64");
  PF_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, PF_ERR_INVALID_ARG, "gemm_rc: lda / ldw must be multiples of 8");
  PF_CHECK((!a.resid || a.ldr % 4 == 0) && (!a.out_x || a.ldx % 4 == 0) && (!a.out_n16 || a.ldn16 % 4 == 0) &&
               (!a.out_n32 || a.ldn32 % 4 == 0),
           PF_ERR_INVALID_ARG, "gemm_rc: leading dimensions must keep 16-byte (8-byte for f16) row alignment");
  PF_CHECK(!a.fsmn_v || (a.fsmn_wT && a.T >= 8 && a.ldv % 4 == 0 && (a.fsmn_k == 11)), PF_ERR_UNSUPPORTED,
           "gemm_rc: the fused FSMN needs k = 11, T >= 8 and an 8-byte aligned V slice");
  PF_CHECK(!a.ln_g == !a.ln_b && (a.ln_g || (!a.out_n16 && !a.out_n32)), PF_ERR_INVALID_ARG,
           "gemm_rc: LayerNorm outputs need gamma and beta");
  PF_CHECK(a.out_x || a.out_n16 || a.out_n32, PF_ERR_INVALID_ARG, "gemm_rc: no output requested");
  RcDev d;
  d.A = a.A; d.W = a.W; d.bias = a.bias; d.resid = a.resid; d.out_x = a.out_x;
  d.fsmn_v = a.fsmn_v; d.fsmn_wT = a.fsmn_wT;
  d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.out_n16 = a.out_n16; d.out_n32 = a.out_n32;
  d.lda = a.lda; d.ldw = a.ldw; d.ldr = a.ldr; d.ldx = a.ldx; d.ldv = a.ldv; d.ldn16 = a.ldn16; d.ldn32 = a.ldn32;
  d.M = a.M; d.K = a.K; d.T = a.T > 0 ? a.T : a.M; d.a_blocked = a.a_blocked;
  d.eps = a.eps;
  static std::mutex init_mu;                         // engines on different devices launch from different threads
  static bool attr_set[64] = {false};
  static int rc_pfd = 0;                             // PF_RC_PFD = 4 | 8: A prefetch touches that many k-steps ahead (deep projections without an FSMN term)
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_rc_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_rc_kernel<11, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_rc_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS + RC_TOUCH));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_rc_kernel<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS + RC_TOUCH));
      if (const char* e = getenv("PF_RC_PFD")) rc_pfd = atoi(e);
      attr_set[dev & 63] = true;
    }
  }
  const dim3 grid((unsigned)cdiv(a.M, RC_BM));
  const int nk = a.K / RC_BK;                        // (the names are the ones the rocprofv3 kernel trace shows)
  if (a.fsmn_v) {
    note_gemm_kernel("gemm_rc_kernel<11, 0>");
    hipLaunchKernelGGL((gemm_rc_kernel<11, 0>), grid, dim3(512), RC_LDS, s, d);
  } else if (rc_pfd >= 8 && nk > 8) {
    note_gemm_kernel("gemm_rc_kernel<0, 8>");
    hipLaunchKernelGGL((gemm_rc_kernel<0, 8>), grid, dim3(512), RC_LDS + RC_TOUCH, s, d);
  } else if (rc_pfd >= 4 && nk > 4) {
    note_gemm_kernel("gemm_rc_kernel<0, 4>");
    hipLaunchKernelGGL((gemm_rc_kernel<0, 4>), grid, dim3(512), RC_LDS + RC_TOUCH, s, d);
  } else {
    note_gemm_kernel("gemm_rc_kernel<0, 0>");
    hipLaunchKernelGGL((gemm_rc_kernel<0, 0>), grid, dim3(512), RC_LDS, s, d);
  }
  PF_HIP(hipGetLastError());
}

}  // namespace pf
