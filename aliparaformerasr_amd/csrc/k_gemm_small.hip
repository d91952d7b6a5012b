// k_gemm_small.hip — GEMM for short inputs (M = B*T <= a few hundred rows: `-method one` on a 5 s file is M = 83
// encoder rows and M = 23 decoder rows; the reference CLI default, AliParaformerAsr.Examples/Program.cs).
//
//   C[M,N] = A[M,K] * W[N,K]^T  (+ bias, q-scale, FSMN memory, residual, ReLU; fp32 and / or f16 result)
//
// (the same MatMul/Gemm (+ Add / Mul / Relu) nodes InferenceSession.Run executes, OfflineProjOfParaformer.cs:68).
//
// Why a separate kernel: at M = 83 the persistent 128 x 128-tile kernel (k_gemm.hip) has 4-16 tiles, so 4-16 of the
// 256 CUs walk 8-32 k-steps and the launch costs one LDS-DMA round trip per k-step — 10 us (K = 512) to 22 us
// (K = 2048) per GEMM, and 465 such launches are the whole 5.7 ms of the 1 x 5 s path.  At this size a launch is a
// chain of dependent memory round trips (~2 us each on this part), so the design rule is: ONE round trip for the
// operands, none for anything else.
//   * A workgroup owns a brick of (up to 128 rows) x 32 columns x KC: it requests every 16-byte piece of the brick
//     before it waits for the first (registers -> swizzled LDS), together with the epilogue operands (bias, residual,
//     FSMN window) it will need at the end, then multiplies out of LDS (4 waves = 4 row blocks of
//     v_mfma_f32_32x32x16_f16, D^T orientation as everywhere else).
//   * K <= 576 (every projection but FFN-down and the CIF conv): KC = K, one brick per output tile, the epilogue runs
//     in the same kernel.  N/32 x ceil(M/128) workgroups (16-512).
//   * K > 576: the K range is cut into S bricks of KC <= 256 whose fp32 partials go to a workspace [S][M][512]; the
//     sum, the epilogue AND the LayerNorm that follows these GEMMs in the graph (FFN-down -> next norm1 / after_norm,
//     decoder FFN-down -> norm2) are one row-wise kernel (small_reduce_kernel) that replaces the LayerNorm launch the
//     pipeline had anyway.  Partials are summed in split order: results are reproducible run to run.
//     (A first version reduced inside the GEMM — write-through partials, arrival counter, last workgroup sums — and
//     measured 12.5 us per launch: three more dependent round trips.)
//     (Taking the LayerNorm in FRONT of a K = 512 projection on load — every workgroup normalising its own copy of
//     the rows — was built and measured: 15.5 us per launch against 9 + 3.9 us for GEMM + LayerNorm kernel, because
//     the fp32 rows double the brick's bytes and the 48 row reductions sit in front of the first MFMA.  Removed.)
//   * The encoder's FSMN memory (11 taps over V, channel-local) is an epilogue term of the attention out-projection:
//     a lane adds sum_j w_j[n] * V[t + j - 5, n] + V[t, n] for its own row and 16 columns.
#include "kernels.h"

#include <cstdlib>
#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

struct SmallDev {
  const half_t* A; int lda;
  const half_t* W; int ldw; const float* bias;
  float* out_f32; half_t* out_f16; const float* resid; const float* add2;
  int ldc32, ldc16, ldr, ld2;
  const half_t* fsmn_v; int ldv; const float* fsmn_wT; int T;      // 11-tap FSMN memory over V (utterances = runs of T rows)
  int relu, scale_cols; float scale;
  int M, N, K, S, tiles_n, tiles_m, bm;                  // bm: brick rows per tile (96 when K = 576 so that it fits LDS)
  float* part;                                           // S > 1: [S][tiles_m * 128][512] fp32 partials, no epilogue here
};

constexpr int SM_BM = 128, SM_BN = 32, SM_PART_LD = 512;

// physical 16-byte chunk of logical chunk c in row `row` (rows of cpr chunks): distinct bank windows for the 16 rows a
// ds_read_b128 lane group touches (MI355X guide, LDS table: 64 banks x 4 B, groups of 16 lanes)
template <int CPR>
__device__ __forceinline__ int sm_swz(int row, int c) {
  return (CPR & 15) ? (c ^ ((row >> 1) & 7)) : (c ^ (row & 15));
}

// wave-wide sum broadcast to every lane: four DPP steps + four SGPR reads (VALU only; ds_bpermute-based shuffles cost
// ~120 cycles each and a LayerNorm is a chain of 12 of them — 15 us for the 24 rows a wave normalises on load)
__device__ __forceinline__ float sm_wave_sum(float v) {
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

// CPR: 16-byte chunks per brick row (KC = 8 * CPR)
template <int CPR>
__global__ __launch_bounds__(256) void gemm_small_kernel(SmallDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KC = CPR * 8, rowb = KC * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tn = b % p.tiles_n; b /= p.tiles_n;
  const int sp = b % p.S;
  const int tm = b / p.S;
  const int m0 = tm * p.bm, n0 = tn * SM_BN, k0 = sp * KC;
  const int rows_here = min(p.bm, p.M - m0);
  const int rb_here = (rows_here + 31) >> 5;                 // 32-row blocks with real rows
  char* lA = smem;
  char* lW = smem + rb_here * 32 * rowb;
  const int lr = lane & 31, lh = lane >> 5;
  const bool active = wave < rb_here;
  const int m = m0 + wave * 32 + lr;                         // the output row of this lane
  const bool row_ok = active && m < p.M;
  const bool epi = p.S == 1;

  // ---- epilogue operands of this lane, requested with the brick (columns n0 + 8*g + 4*lh + e)
  float4 ebias[4], eres[4], eadd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n0 + 8 * g + 4 * lh;
    const bool ok = epi && row_ok && n + 3 < p.N;
    ebias[g] = (ok && p.bias) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    eres[g] = (ok && p.resid) ? *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    eadd[g] = (ok && p.add2) ? *reinterpret_cast<const float4*>(p.add2 + (size_t)m * p.ld2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // FSMN window of this lane: 11 rows x 16 columns of V (f16); taps are read in the epilogue (L2-resident, 22 KB)
  h4 fv[11][4];
  const bool do_fsmn = epi && p.fsmn_v != nullptr;
  if (do_fsmn) {
    const int mm = row_ok ? m : (p.M - 1);
    const int t = mm % p.T;
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int tt = t + j - 5;
      const bool ok = tt >= 0 && tt < p.T;
      const half_t* vr = p.fsmn_v + (size_t)(ok ? mm + j - 5 : mm) * p.ldv + n0 + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const h4 x = *reinterpret_cast<const h4*>(vr + 8 * g);
        fv[j][g] = ok ? x : h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      }
    }
  }

  // ---- the brick, one shot: every piece is requested before the first one is awaited (rows past M / N are clamped:
  // they only feed result rows / columns nobody stores)
  {
    constexpr int MAXC = ((SM_BM + SM_BN) * CPR + 255) / 256;
    const int nA = rb_here * 32 * CPR, total = nA + SM_BN * CPR;
    h8 v[MAXC];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int i = tid + 256 * u;
      if (i < total) {
        const bool isA = i < nA;
        const int j = isA ? i : i - nA;
        const int row = j / CPR, c = j - row * CPR;
        const half_t* src = isA ? p.A + (size_t)min(m0 + row, p.M - 1) * p.lda : p.W + (size_t)min(n0 + row, p.N - 1) * p.ldw;
        v[u] = *reinterpret_cast<const h8*>(src + k0 + c * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int i = tid + 256 * u;
      if (i < total) {
        const bool isA = i < nA;
        const int j = isA ? i : i - nA;
        const int row = j / CPR, c = j - row * CPR;
        *reinterpret_cast<h8*>((isA ? lA : lW) + row * rowb + (sm_swz<CPR>(row, c) << 4)) = v[u];
      }
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- wave = one 32-row block; D^T = W_tile * X_tile^T: a lane owns output row (lane & 31) and 4 consecutive
  // columns per register quad (column 8*g + 4*(lane >> 5) + e).  Two accumulators break the MFMA dependency chain.
  f16x acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  {
    const int ra = wave * 32 + lr;
    const char* pa = lA + ra * rowb;
    const char* pw = lW + lr * rowb;
    constexpr int NK = KC / 16;
#pragma unroll 4
    for (int kk = 0; kk + 1 < NK; kk += 2) {
      const int c = 2 * kk + lh;
      const h8 af0 = *reinterpret_cast<const h8*>(pa + (sm_swz<CPR>(ra, c) << 4));
      const h8 bf0 = *reinterpret_cast<const h8*>(pw + (sm_swz<CPR>(lr, c) << 4));
      const h8 af1 = *reinterpret_cast<const h8*>(pa + (sm_swz<CPR>(ra, c + 2) << 4));
      const h8 bf1 = *reinterpret_cast<const h8*>(pw + (sm_swz<CPR>(lr, c + 2) << 4));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf0, af0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf1, af1, acc1, 0, 0, 0);
    }
    if (NK & 1) {
      const int c = 2 * (NK - 1) + lh;
      const h8 af0 = *reinterpret_cast<const h8*>(pa + (sm_swz<CPR>(ra, c) << 4));
      const h8 bf0 = *reinterpret_cast<const h8*>(pw + (sm_swz<CPR>(lr, c) << 4));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf0, af0, acc0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) acc0[e] += acc1[e];

  if (!epi) {                                                // partial [sp][m][n], rows padded to the tile
    float* pr = p.part + ((size_t)sp * p.tiles_m * p.bm + m) * SM_PART_LD + n0 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(pr + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
    return;
  }
  if (!row_ok) return;
  // ---- epilogue (order as k_gemm.hip: bias, q-scale, FSMN memory / addend, residual, ReLU)
  const float lo = p.relu ? 0.f : -INFINITY;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n0 + 8 * g + 4 * lh;
    if (n >= p.N) continue;
    float v[4] = {acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
    if (n + 3 < p.N) {
      v[0] += ebias[g].x; v[1] += ebias[g].y; v[2] += ebias[g].z; v[3] += ebias[g].w;
      if (n < p.scale_cols) { v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale; }
      if (do_fsmn) {
        float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 11; ++j) {
          const float4 w4 = *reinterpret_cast<const float4*>(p.fsmn_wT + (size_t)j * 512 + n);
          f[0] += w4.x * (float)fv[j][g][0]; f[1] += w4.y * (float)fv[j][g][1];
          f[2] += w4.z * (float)fv[j][g][2]; f[3] += w4.w * (float)fv[j][g][3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += f[e] + (float)fv[5][g][e];
      }
      v[0] += eadd[g].x; v[1] += eadd[g].y; v[2] += eadd[g].z; v[3] += eadd[g].w;
      v[0] += eres[g].x; v[1] += eres[g].y; v[2] += eres[g].z; v[3] += eres[g].w;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
      if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc32 + n) = make_float4(v[0], v[1], v[2], v[3]);
      if (p.out_f16) *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    } else {                                                 // ragged last columns (N % 4 != 0): scalar, operands re-read
      for (int e = 0; e < 4 && n + e < p.N; ++e) {
        float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
        if (n + e < p.scale_cols) x *= p.scale;
        if (p.add2) x += p.add2[(size_t)m * p.ld2 + n + e];
        if (p.resid) x += p.resid[(size_t)m * p.ldr + n + e];
        x = fmaxf(x, lo);
        if (p.out_f32) p.out_f32[(size_t)m * p.ldc32 + n + e] = x;
        if (p.out_f16) p.out_f16[(size_t)m * p.ldc16 + n + e] = (half_t)x;
      }
    }
  }
}

// Sum of the S partials of a split GEMM (N = 512) in split order + bias + addend + residual [+ ReLU] -> fp32 / f16 result,
// and optionally LayerNorm of that result -> f16 / fp32 (the LayerNorm launch that followed the GEMM in the pipeline).
// One wave per row; a lane holds columns 4*lane..+3 and 256+4*lane..+3.
struct ReduceDev {
  const float* part; int S; size_t slab;                   // slab = floats per split
  const float* bias; const float* resid; int ldr; const float* add2; int ld2; int relu;
  float* out_f32; int ldc32; half_t* out_f16; int ldc16;
  const float* ln_g; const float* ln_b; half_t* n16; int ldn16; float* n32; int ldn32;
  int M;
};

__global__ __launch_bounds__(256) void small_reduce_kernel(ReduceDev p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int c0 = 4 * lane, c1 = 256 + 4 * lane;
  const float* pr = p.part + (size_t)m * SM_PART_LD;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
  constexpr int SB = 8;
  for (int s0 = 0; s0 < p.S; s0 += SB) {
    float4 va[SB], vc[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int s = min(s0 + u, p.S - 1);
      va[u] = *reinterpret_cast<const float4*>(pr + (size_t)s * p.slab + c0);
      vc[u] = *reinterpret_cast<const float4*>(pr + (size_t)s * p.slab + c1);
    }
#pragma unroll
    for (int u = 0; u < SB; ++u)
      if (s0 + u < p.S) {
        a.x += va[u].x; a.y += va[u].y; a.z += va[u].z; a.w += va[u].w;
        c.x += vc[u].x; c.y += vc[u].y; c.z += vc[u].z; c.w += vc[u].w;
      }
  }
  auto add4 = [](float4& x, const float* q) __attribute__((always_inline)) {
    const float4 y = *reinterpret_cast<const float4*>(q);
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
  };
  if (p.bias) { add4(a, p.bias + c0); add4(c, p.bias + c1); }
  if (p.add2) { add4(a, p.add2 + (size_t)m * p.ld2 + c0); add4(c, p.add2 + (size_t)m * p.ld2 + c1); }
  if (p.resid) { add4(a, p.resid + (size_t)m * p.ldr + c0); add4(c, p.resid + (size_t)m * p.ldr + c1); }
  if (p.relu) {
    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
  }
  if (p.out_f32) {
    *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc32 + c0) = a;
    *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc32 + c1) = c;
  }
  if (p.out_f16) {
    *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + c0) = h4{(half_t)a.x, (half_t)a.y, (half_t)a.z, (half_t)a.w};
    *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + c1) = h4{(half_t)c.x, (half_t)c.y, (half_t)c.z, (half_t)c.w};
  }
  if (!p.ln_g) return;
  const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.x)));
  a.x -= x0; a.y -= x0; a.z -= x0; a.w -= x0; c.x -= x0; c.y -= x0; c.z -= x0; c.w -= x0;
  const float mean = sm_wave_sum(((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w))) * (1.0f / 512.0f);
  a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean; c.x -= mean; c.y -= mean; c.z -= mean; c.w -= mean;
  const float var = sm_wave_sum(((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w))) * (1.0f / 512.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-12f);
  const float4 g0 = *reinterpret_cast<const float4*>(p.ln_g + c0), g1 = *reinterpret_cast<const float4*>(p.ln_g + c1);
  const float4 e0 = *reinterpret_cast<const float4*>(p.ln_b + c0), e1 = *reinterpret_cast<const float4*>(p.ln_b + c1);
  const float4 ya = make_float4(a.x * rstd * g0.x + e0.x, a.y * rstd * g0.y + e0.y, a.z * rstd * g0.z + e0.z, a.w * rstd * g0.w + e0.w);
  const float4 yc = make_float4(c.x * rstd * g1.x + e1.x, c.y * rstd * g1.y + e1.y, c.z * rstd * g1.z + e1.z, c.w * rstd * g1.w + e1.w);
  if (p.n32) {
    *reinterpret_cast<float4*>(p.n32 + (size_t)m * p.ldn32 + c0) = ya;
    *reinterpret_cast<float4*>(p.n32 + (size_t)m * p.ldn32 + c1) = yc;
  }
  if (p.n16) {
    *reinterpret_cast<h4*>(p.n16 + (size_t)m * p.ldn16 + c0) = h4{(half_t)ya.x, (half_t)ya.y, (half_t)ya.z, (half_t)ya.w};
    *reinterpret_cast<h4*>(p.n16 + (size_t)m * p.ldn16 + c1) = h4{(half_t)yc.x, (half_t)yc.y, (half_t)yc.z, (half_t)yc.w};
  }
}

// rows up to which the pipeline prefers this kernel (PF_SMALL_M overrides; 0 disables)
int gemm_small_max_rows() {
  static const int v = std::min(env_int("PF_SMALL_M", 512), 512);   // workspace rows; measured:
  // M = 664 (8 x 5 s) ties with the persistent kernel (5.51 vs 5.44 ms), M = 1000 (2 x 30 s) loses (6.56 vs 6.06 ms)
  return v;
}
size_t gemm_small_ws_bytes() { return (size_t)8 * 512 * SM_PART_LD * 4; }   // 8 splits x 512 rows x 512 columns fp32 = 8 MiB

static int small_bm(int K) { return K > 512 && K <= 576 ? 96 : SM_BM; }   // (bm + 32) * K * 2 bytes of LDS <= 160 KiB
static int small_split(const GemmSmallArgs& a) {
  if (a.K <= 576 && a.K % 64 == 0) return 1;
  const int nkb = a.K / 64;
  for (int s = 2; s <= nkb; ++s)
    if (nkb % s == 0 && (nkb / s) * 64 <= 256) return s;
  return 0;
}

bool gemm_small_applicable(const GemmSmallArgs& a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % 64 != 0) return false;
  const int S = small_split(a);
  if (S == 0) return false;
  if (!a.A) return false;
  if (a.fsmn_v && (S != 1 || a.fsmn_k != 11 || a.N != 512 || a.T <= 0)) return false;
  if (S == 1) return !a.post_ln_g;                           // a LayerNorm behind the GEMM exists only on the split path
  if (a.N != 512 || !a.ws) return false;                     // split path: row-wise reduction over N = 512
  if (a.scale_cols > 0) return false;
  return (size_t)S * cdiv(a.M, SM_BM) * SM_BM * SM_PART_LD * 4 <= gemm_small_ws_bytes();   // split bricks are 128 rows
}

template <int CPR>
static void small_launch(hipStream_t s, const SmallDev& d, int rows_alloc) {
  static std::mutex mu;
  static bool attr[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_small_kernel<CPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr[dev & 63] = true;
    }
  }
  const int lds = (rows_alloc + SM_BN) * CPR * 16;
  note_gemm_kernel("gemm_small_kernel");
  hipLaunchKernelGGL((gemm_small_kernel<CPR>), dim3((unsigned)(d.tiles_n * d.tiles_m * d.S)), dim3(256), lds, s, d);
  PF_HIP(hipGetLastError());
}

void launch_gemm_small(hipStream_t s, const GemmSmallArgs& a) {
  PF_CHECK(gemm_small_applicable(a), PF_ERR_INVALID_ARG, "gemm_small: the short-input kernel does not apply to this problem");
  SmallDev d{};
  d.A = a.A; d.lda = a.lda;
  d.W = a.W; d.ldw = a.ldw; d.bias = a.bias;
  d.out_f32 = a.out_f32; d.out_f16 = a.out_f16; d.resid = a.resid; d.add2 = a.add2;
  d.ldc32 = a.ldc32; d.ldc16 = a.ldc16; d.ldr = a.ldr; d.ld2 = a.ld2;
  d.fsmn_v = a.fsmn_v; d.ldv = a.ldv; d.fsmn_wT = a.fsmn_wT; d.T = a.T > 0 ? a.T : a.M;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.S = small_split(a);
  d.bm = d.S == 1 ? small_bm(a.K) : SM_BM;
  d.tiles_n = cdiv(a.N, SM_BN); d.tiles_m = cdiv(a.M, d.bm);
  d.part = a.ws;
  const int rows_alloc = a.M >= d.bm ? d.bm : (int)round_up(a.M, 32);   // LDS rows of the A brick (largest tile)
  const int kc = a.K / d.S;
  if (d.S > 1) {                                             // partials only; bias / residual / LayerNorm in the reduction
    d.bias = nullptr; d.resid = nullptr; d.add2 = nullptr; d.out_f32 = nullptr; d.out_f16 = nullptr; d.relu = 0;
  }
  switch (kc / 64) {
    case 1: small_launch<8>(s, d, rows_alloc); break;
    case 2: small_launch<16>(s, d, rows_alloc); break;
    case 3: small_launch<24>(s, d, rows_alloc); break;
    case 4: small_launch<32>(s, d, rows_alloc); break;
    case 5: small_launch<40>(s, d, rows_alloc); break;
    case 6: small_launch<48>(s, d, rows_alloc); break;
    case 7: small_launch<56>(s, d, rows_alloc); break;
    case 8: small_launch<64>(s, d, rows_alloc); break;
    default: small_launch<72>(s, d, rows_alloc); break;
  }
  if (d.S > 1) {
    ReduceDev r{};
    r.part = a.ws; r.S = d.S; r.slab = (size_t)d.tiles_m * d.bm * SM_PART_LD;
    r.bias = a.bias; r.resid = a.resid; r.ldr = a.ldr; r.add2 = a.add2; r.ld2 = a.ld2; r.relu = a.relu;
    r.out_f32 = a.out_f32; r.ldc32 = a.ldc32; r.out_f16 = a.out_f16; r.ldc16 = a.ldc16;
    r.ln_g = a.post_ln_g; r.ln_b = a.post_ln_b; r.n16 = a.post_n16; r.ldn16 = a.ldn16; r.n32 = a.post_n32; r.ldn32 = a.ldn32;
    r.M = a.M;
    hipLaunchKernelGGL(small_reduce_kernel, dim3((unsigned)cdiv(a.M, 4)), dim3(256), 0, s, r);
    PF_HIP(hipGetLastError());
  }
}

}  // namespace pf
