// k_fp32.hip — the fp32 PARITY mode (pf_engine_config.math_mode = 1; SURVEY.md §7 "hard parts", §8c plan): every
// product of the graph InferenceSession.Run executes (AliParaformerAsr/OfflineProjOfParaformer.cs:68) is formed from
// fp32 operands on the fp32 matrix path — v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation, bit-for-bit an
// fmaf chain in k order (MI355X guide §3) — so the only differences left against an fp32 CPU run of the same graph are
// accumulation order.  1/16 of the f16 MFMA rate by construction and deliberately simple (LDS-staged 64 x 64 tiles, no
// software pipelining): this mode exists to check token identity against fp32 references, not to be benchmarked.
#include "kernels.h"
#include "exact.h"

namespace pf {

typedef float f16x __attribute__((ext_vector_type(16)));

struct G32Dev {
  const float* A; const float* W; const float* bias; const float* resid; float* out;
  int lda, ldw, ldr, ldc, M, N, K, relu, scale_cols;
  float scale;
};

// C[M,N] = A[M,K] W[N,K]^T (+ bias) (* scale on columns < scale_cols) (+ resid) (ReLU).  64 x 64 tile per 256-thread
// workgroup, wave w owns the 32 x 32 block (w >> 1, w & 1); K in steps of 32 staged through LDS (rows padded to 33).
__global__ __launch_bounds__(256) void gemm_f32_kernel(G32Dev p) {
  __shared__ float As[64][33], Ws[64][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  f16x acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    for (int e = tid; e < 64 * 32; e += 256) {
      const int r = e >> 5, c = e & 31;
      const int m = m0 + r, n = n0 + r, k = k0 + c;
      As[r][c] = (m < p.M && k < p.K) ? p.A[(size_t)m * p.lda + k] : 0.f;
      Ws[r][c] = (n < p.N && k < p.K) ? p.W[(size_t)n * p.ldw + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = As[wr + (lane & 31)][kk + (lane >> 5)];
      const float b = Ws[wc + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wc + (lane & 31);
  if (n >= p.N) return;
  const float bv = p.bias ? p.bias[n] : 0.f;
  const float sc = n < p.scale_cols ? p.scale : 1.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + wr + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    if (m >= p.M) continue;
    float v = (acc[reg] + bv) * sc;
    if (p.resid) v += p.resid[(size_t)m * p.ldr + n];
    if (p.relu) v = fmaxf(v, 0.f);
    p.out[(size_t)m * p.ldc + n] = v;
  }
}

void launch_gemm_f32(hipStream_t s, const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K,
                     float* out, int ldc, const float* resid, int ldr, bool relu, int scale_cols, float scale) {
  if (M <= 0 || N <= 0) return;
  G32Dev d{A, W, bias, resid, out, lda, ldw, ldr, ldc, M, N, K, relu ? 1 : 0, scale_cols, scale_cols > 0 ? scale : 1.f};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64)), dim3(256), 0, s, d);
  PF_HIP(hipGetLastError());
}

// softmax(q k^T) v for one (query, head) per workgroup, head dim 128, q pre-scaled; element strides as AttnArgs.
__global__ __launch_bounds__(128) void attn_f32_kernel(const float* __restrict__ q, int64_t q_bs, int q_rs, const float* __restrict__ k,
                                                       int64_t k_bs, int k_rs, const float* __restrict__ v, int64_t v_bs, int v_rs,
                                                       float* __restrict__ o, int64_t o_bs, int o_rs, int H, int Lq, int Lk) {
  extern __shared__ float sm[];                    // [128] q row, [Lk] scores, [4] reductions
  float* qs = sm; float* sc = sm + 128; float* red = sm + 128 + Lk;
  const int iq = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H, tid = threadIdx.x;
  qs[tid] = q[(size_t)b * q_bs + (size_t)iq * q_rs + h * 128 + tid];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < Lk; j += 128) {
    const float* kr = k + (size_t)b * k_bs + (size_t)j * k_rs + h * 128;
    float s = 0.f;
    for (int d = 0; d < 128; ++d) s += qs[d] * kr[d];
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(red[0], red[1]);
  float sum = 0.f;
  for (int j = tid; j < Lk; j += 128) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[2 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[2] + red[3]);
  const float* vc = v + (size_t)b * v_bs + h * 128 + tid;
  float acc = 0.f;
  for (int j = 0; j < Lk; ++j) acc += (sc[j] * inv) * vc[(size_t)j * v_rs];
  o[(size_t)b * o_bs + (size_t)iq * o_rs + h * 128 + tid] = acc;
}

// The same on the fp32 matrix path (round 5; the kernel above took 135 of the 240 ms of an fp32-mode step): flash form, one
// workgroup = 128 queries of one (utterance, head), 4 waves x 32 queries, key tiles of 32 staged through LDS as fp32 rows
// (row stride 132 floats: the fragment reads are conflict-free).
//   S^T[32 keys x 32 q] = K Q^T : 64 x v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation); a lane owns one query
//     column, Q lives in 64 registers for the whole launch, a K fragment is one ds_read_b128 = four k-steps (the k index of an
//     MFMA is a summation index: lane (row, kh) takes d = 8 g + 4 kh + e for step 4 g + e, on both operands);
//   online softmax in fp32 (expf, running max / sum per lane, one lane^32 exchange per tile);
//   O^T[128 d x 32 q] += V^T P^T : 64 MFMAs, the P operand of step 4 g + j IS the lane's own S register 4 g + j (D layout:
//     keys 8 g + 4 kh + j), V^T fragments are ds_read_b32 rows of the V tile.
// Invalid keys of the tail tile score -inf and read the last valid V row (0 x NaN of a stale row would poison O).
constexpr int AF_KT = 32, AF_LDK = 132;
__global__ __launch_bounds__(256) void attn_f32_mfma_kernel(const float* __restrict__ q, int64_t q_bs, int q_rs, const float* __restrict__ k,
                                                            int64_t k_bs, int k_rs, const float* __restrict__ v, int64_t v_bs, int v_rs,
                                                            float* __restrict__ o, int64_t o_bs, int o_rs, int H, int Lq, int Lk,
                                                            half_t* __restrict__ opair = nullptr, int p_lo = 0) {
  __shared__ __attribute__((aligned(16))) float Ks[AF_KT * AF_LDK], Vs[AF_KT * AF_LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = min(q0 + l31, Lq - 1);                      // (rows past Lq compute on a copy of the last row; never stored)
  const float* qp = q + (size_t)b * q_bs + (size_t)qrow * q_rs + h * 128;
  float qf[64];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const float4 t = *reinterpret_cast<const float4*>(qp + 8 * g + 4 * kh);
    qf[4 * g + 0] = t.x; qf[4 * g + 1] = t.y; qf[4 * g + 2] = t.z; qf[4 * g + 3] = t.w;
  }
  f16x oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float* kb = k + (size_t)b * k_bs + h * 128;
  const float* vb = v + (size_t)b * v_bs + h * 128;
  for (int k0 = 0; k0 < Lk; k0 += AF_KT) {
    __syncthreads();                                           // everybody has finished with the previous tile
    // 32 keys x 128 floats of K and of V: 1024 float4 each, 4 + 4 per thread
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + it * 256, r = e >> 5, c4 = e & 31;
      const int kr = min(k0 + r, Lk - 1);
      *reinterpret_cast<float4*>(&Ks[r * AF_LDK + 4 * c4]) = *reinterpret_cast<const float4*>(kb + (size_t)kr * k_rs + 4 * c4);
      *reinterpret_cast<float4*>(&Vs[r * AF_LDK + 4 * c4]) = *reinterpret_cast<const float4*>(vb + (size_t)kr * v_rs + 4 * c4);
    }
    __syncthreads();
    f16x sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float4 kf = *reinterpret_cast<const float4*>(&Ks[l31 * AF_LDK + 8 * g + 4 * kh]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * g + 0], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * g + 1], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * g + 2], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * g + 3], sacc, 0, 0, 0);
    }
    // this lane: query l31, keys k0 + 8 g + 4 kh + j in register 4 g + j
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = k0 + 8 * (e >> 2) + 4 * kh + (e & 3);
      if (key >= Lk) sacc[e] = -INFINITY;
      mt = fmaxf(mt, sacc[e]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);                      // finite: a tile holds at least one valid key
    const float alpha = expf(m_run - m_new);                   // first tile: exp(-inf) = 0
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sacc[e] = expf(sacc[e] - m_new); ls += sacc[e]; }
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float vf = Vs[(8 * (e >> 2) + 4 * kh + (e & 3)) * AF_LDK + d * 32 + l31];
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sacc[e], oacc[d], 0, 0, 0);
      }
  }
  if (q0 + l31 >= Lq) return;
  const float inv = 1.0f / l_run;
  if (opair) {
    // math_mode 3: the context is only the A operand of the out-projection — it leaves as that product's (hi | lo') pair
    // (o_bs / o_rs then index the pair matrix, p_lo = column offset of the lo' half)
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    half_t* pp = opair + (size_t)b * o_bs + (size_t)(q0 + l31) * o_rs + h * 128;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = oacc[d][4 * g + 0] * inv, v1 = oacc[d][4 * g + 1] * inv, v2 = oacc[d][4 * g + 2] * inv, v3 = oacc[d][4 * g + 3] * inv;
        const h4 hv = h4{(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
        *reinterpret_cast<h4*>(pp + d * 32 + 8 * g + 4 * kh) = hv;
        *reinterpret_cast<h4*>(pp + p_lo + d * 32 + 8 * g + 4 * kh) =
            h4{(half_t)((v0 - (float)hv[0]) * 2048.f), (half_t)((v1 - (float)hv[1]) * 2048.f), (half_t)((v2 - (float)hv[2]) * 2048.f),
               (half_t)((v3 - (float)hv[3]) * 2048.f)};
      }
    return;
  }
  float* op = o + (size_t)b * o_bs + (size_t)(q0 + l31) * o_rs + h * 128;
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(op + d * 32 + 8 * g + 4 * kh) =
          make_float4(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
}

// The MFMA form with the context written as an x3 operand pair (hi at pair[b * p_bs + q * p_rs + h * 128 + d], lo' p_lo columns on).
// Returns false when the MFMA form does not apply (alignment): the caller then keeps the fp32 result + split.
bool launch_attention_f32_pair(hipStream_t s, const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs,
                               const float* v, int64_t v_bs, int v_rs, half_t* pair, int64_t p_bs, int p_rs, int p_lo, int B, int H, int Lq, int Lk) {
  if (B * H == 0 || Lq == 0) return true;
  const bool aligned = ((q_rs | k_rs | v_rs | p_rs | p_lo) % 4 == 0) && (q_bs % 4 == 0) && (k_bs % 4 == 0) && (v_bs % 4 == 0) && (p_bs % 4 == 0) &&
                       ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0) && (((uintptr_t)pair & 7) == 0);
  if (!aligned || Lk <= 0) return false;
  hipLaunchKernelGGL(attn_f32_mfma_kernel, dim3((unsigned)cdiv(Lq, 128), (unsigned)(B * H)), dim3(256), 0, s, q, q_bs, q_rs, k, k_bs, k_rs,
                     v, v_bs, v_rs, (float*)nullptr, p_bs, p_rs, H, Lq, Lk, pair, p_lo);
  PF_HIP(hipGetLastError());
  return true;
}

void launch_attention_f32(hipStream_t s, const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs,
                          const float* v, int64_t v_bs, int v_rs, float* o, int64_t o_bs, int o_rs, int B, int H, int Lq, int Lk) {
  if (B == 0 || Lq == 0 || Lk == 0) return;
  static const bool naive = [] { const char* e = getenv("PF_ATTN_F32_NAIVE"); return e && e[0] == '1'; }();   // A/B: the one-query-per-workgroup kernel
  const bool aligned = ((q_rs | k_rs | v_rs | o_rs) % 4 == 0) && (q_bs % 4 == 0) && (k_bs % 4 == 0) && (v_bs % 4 == 0) && (o_bs % 4 == 0) &&
                       ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0);
  if (!naive && aligned) {
    hipLaunchKernelGGL(attn_f32_mfma_kernel, dim3((unsigned)cdiv(Lq, 128), (unsigned)(B * H)), dim3(256), 0, s, q, q_bs, q_rs, k, k_bs, k_rs,
                       v, v_bs, v_rs, o, o_bs, o_rs, H, Lq, Lk);
    PF_HIP(hipGetLastError());
    return;
  }
  PF_CHECK((size_t)(128 + Lk + 4) * 4 <= 64 * 1024, PF_ERR_UNSUPPORTED, "fp32 mode: more than ~16000 keys per utterance");
  hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)Lq, (unsigned)(B * H)), dim3(128), (size_t)(128 + Lk + 4) * 4, s, q, q_bs, q_rs,
                     k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, H, Lq, Lk);
  PF_HIP(hipGetLastError());
}

// The same flash attention on the f16 matrix cores with "x3" operands (math_mode 3): every fp32 operand as the pair
// hi = f16(x), lo' = f16((x - hi) * 2^11) (22 mantissa bits) and every product as hi hi + 2^-11 (hi lo' + lo' hi), the cross terms in
// an accumulator of their own.  S^T = K Q^T: 8 k-steps x 3 v_mfma_f32_32x32x16_f16 per 32 keys x 32 queries instead of 64 fp32
// MFMAs at 1/16 of the rate; O^T = V^T P^T: 4 d-blocks x 2 k-steps x 3.  Q pairs live in registers, K pairs in LDS as rows, V
// pairs in LDS TRANSPOSED (keys contiguous per d: the k index of an MFMA operand is contiguous per lane); the P operand of k-step
// t is the lane's own probabilities 8 t .. 8 t + 7 (D layout keys 16 t + 8 (e >> 2) + 4 kh + (e & 3)), V^T fragments follow that
// key order (two 8-byte reads).  fp32 softmax as above.
typedef _Float16 x3h8 __attribute__((ext_vector_type(8)));
typedef _Float16 x3h4 __attribute__((ext_vector_type(4)));
constexpr int AX_LDK = 136, AX_LDV = 36;                       // halves per K row (272 B) / per V^T row (72 B): conflict-free reads
__device__ __forceinline__ void x3_split(float v, half_t& hi, half_t& lo) {
  hi = (half_t)v;
  lo = (half_t)((v - (float)hi) * 2048.0f);
}
// SF32: the SCORES on the fp32 matrix path (exact fp32 products: what is exponentiated), only P V with x3 operands
template <bool SF32>
__global__ __launch_bounds__(256) void attn_x3_kernel(const float* __restrict__ q, int64_t q_bs, int q_rs, const float* __restrict__ k,
                                                      int64_t k_bs, int k_rs, const float* __restrict__ v, int64_t v_bs, int v_rs,
                                                      float* __restrict__ o, int64_t o_bs, int o_rs, int H, int Lq, int Lk) {
  __shared__ __attribute__((aligned(16))) half_t Kh[SF32 ? 8 : 32 * AX_LDK], Kl[SF32 ? 8 : 32 * AX_LDK], Vh[128 * AX_LDV], Vl[128 * AX_LDV];
  __shared__ __attribute__((aligned(16))) float Kf[SF32 ? AF_KT * AF_LDK : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = min(q0 + l31, Lq - 1);
  const float* qp = q + (size_t)b * q_bs + (size_t)qrow * q_rs + h * 128;
  x3h8 qh[SF32 ? 1 : 8], ql[SF32 ? 1 : 8];                      // B operand of k-step s: d = 16 s + 8 kh + 0..7
  float qf[SF32 ? 64 : 1];                                      // SF32: fp32 k-step 4 g + e <-> d = 8 g + 4 kh + e (attn_f32_mfma_kernel)
  if constexpr (SF32) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 8 * g + 4 * kh);
      qf[4 * g + 0] = t.x; qf[4 * g + 1] = t.y; qf[4 * g + 2] = t.z; qf[4 * g + 3] = t.w;
    }
  } else {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * kh), t1 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * kh + 4);
    const float f[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) { half_t a, c; x3_split(f[e], a, c); qh[s][e] = a; ql[s][e] = c; }
  }
  }
  f16x om[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) om[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float* kb = k + (size_t)b * k_bs + h * 128;
  const float* vb = v + (size_t)b * v_bs + h * 128;
  for (int k0 = 0; k0 < Lk; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + it * 256, r = e >> 5, c4 = e & 31;
      const int kr = min(k0 + r, Lk - 1);
      const float4 kf = *reinterpret_cast<const float4*>(kb + (size_t)kr * k_rs + 4 * c4);
      const float4 vf = *reinterpret_cast<const float4*>(vb + (size_t)kr * v_rs + 4 * c4);
      if constexpr (SF32) {
        *reinterpret_cast<float4*>(&Kf[r * AF_LDK + 4 * c4]) = kf;
      } else {
        x3h4 a, c;
        { half_t x, y; x3_split(kf.x, x, y); a[0] = x; c[0] = y; x3_split(kf.y, x, y); a[1] = x; c[1] = y;
          x3_split(kf.z, x, y); a[2] = x; c[2] = y; x3_split(kf.w, x, y); a[3] = x; c[3] = y; }
        *reinterpret_cast<x3h4*>(&Kh[r * AX_LDK + 4 * c4]) = a;
        *reinterpret_cast<x3h4*>(&Kl[r * AX_LDK + 4 * c4]) = c;
      }
      const float vv[4] = {vf.x, vf.y, vf.z, vf.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { half_t x, y; x3_split(vv[j], x, y); Vh[(4 * c4 + j) * AX_LDV + r] = x; Vl[(4 * c4 + j) * AX_LDV + r] = y; }
    }
    __syncthreads();
    f16x sm, sc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sm[e] = 0.f; sc[e] = 0.f; }
    if constexpr (SF32) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float4 kf = *reinterpret_cast<const float4*>(&Kf[l31 * AF_LDK + 8 * g + 4 * kh]);
        sm = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * g + 0], sm, 0, 0, 0);
        sm = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * g + 1], sm, 0, 0, 0);
        sm = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * g + 2], sm, 0, 0, 0);
        sm = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * g + 3], sm, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const x3h8 ah = *reinterpret_cast<const x3h8*>(&Kh[l31 * AX_LDK + 16 * s + 8 * kh]);
      const x3h8 al = *reinterpret_cast<const x3h8*>(&Kl[l31 * AX_LDK + 16 * s + 8 * kh]);
      sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], sm, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], sc, 0, 0, 0);
    }
    }
    float p[16];
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = k0 + 8 * (e >> 2) + 4 * kh + (e & 3);
      p[e] = key < Lk ? sm[e] + sc[e] * (1.0f / 2048.0f) : -INFINITY;
      mt = fmaxf(mt, p[e]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { p[e] = expf(p[e] - m_new); ls += p[e]; }
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
    x3h8 ph[2], pl[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { half_t a, c; x3_split(p[e], a, c); ph[e >> 3][e & 7] = a; pl[e >> 3][e & 7] = c; }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      f16x oc;
#pragma unroll
      for (int e = 0; e < 16; ++e) { om[d][e] *= alpha; oc[e] = 0.f; }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // A operand: lane (d = 32 d + l31, kh): keys 16 t + 4 kh + 0..3 and 16 t + 8 + 4 kh + 0..3
        const half_t* vhp = &Vh[(d * 32 + l31) * AX_LDV + 16 * t + 4 * kh];
        const half_t* vlp = &Vl[(d * 32 + l31) * AX_LDV + 16 * t + 4 * kh];
        const x3h4 h0 = *reinterpret_cast<const x3h4*>(vhp), h1 = *reinterpret_cast<const x3h4*>(vhp + 8);
        const x3h4 l0 = *reinterpret_cast<const x3h4*>(vlp), l1 = *reinterpret_cast<const x3h4*>(vlp + 8);
        const x3h8 vhf = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), vlf = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        om[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhf, ph[t], om[d], 0, 0, 0);
        oc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhf, pl[t], oc, 0, 0, 0);
        oc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlf, ph[t], oc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) om[d][e] += oc[e] * (1.0f / 2048.0f);
    }
  }
  if (q0 + l31 >= Lq) return;
  const float inv = 1.0f / l_run;
  float* op = o + (size_t)b * o_bs + (size_t)(q0 + l31) * o_rs + h * 128;
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(op + d * 32 + 8 * g + 4 * kh) =
          make_float4(om[d][4 * g + 0] * inv, om[d][4 * g + 1] * inv, om[d][4 * g + 2] * inv, om[d][4 * g + 3] * inv);
}

void launch_attention_x3(hipStream_t s, const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs,
                         const float* v, int64_t v_bs, int v_rs, float* o, int64_t o_bs, int o_rs, int B, int H, int Lq, int Lk, bool scores_f32) {
  if (B == 0 || Lq == 0 || Lk == 0) return;
  const bool aligned = ((q_rs | k_rs | v_rs | o_rs) % 4 == 0) && (q_bs % 4 == 0) && (k_bs % 4 == 0) && (v_bs % 4 == 0) && (o_bs % 4 == 0) &&
                       ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0);
  if (!aligned) { launch_attention_f32(s, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk); return; }
  if (scores_f32)
    hipLaunchKernelGGL(attn_x3_kernel<true>, dim3((unsigned)cdiv(Lq, 128), (unsigned)(B * H)), dim3(256), 0, s, q, q_bs, q_rs, k, k_bs, k_rs,
                       v, v_bs, v_rs, o, o_bs, o_rs, H, Lq, Lk);
  else
    hipLaunchKernelGGL(attn_x3_kernel<false>, dim3((unsigned)cdiv(Lq, 128), (unsigned)(B * H)), dim3(256), 0, s, q, q_bs, q_rs, k, k_bs, k_rs,
                       v, v_bs, v_rs, o, o_bs, o_rs, H, Lq, Lk);
  PF_HIP(hipGetLastError());
}

// ---- "x3" products (math_mode 3): an fp32 value as TWO f16 numbers, hi = f16(x) and lo' = f16((x - hi) * 2^11) — 22 bits of
// mantissa, the low part kept in the normal range by the scaling — so that x y = hi_x hi_y + 2^-11 (hi_x lo'_y + lo'_x hi_y)
// up to 2^-22 relative: three f16 MFMA products (16x the fp32 matrix rate each) with fp32 accumulation.
// Row r of x [rows, K] (fp32, row stride ldx) -> out[r, 0:Kp] | out[r, Kp:2 Kp] = hi | lo' (swap = 0) or lo' | hi (swap = 1);
// columns K..Kp-1 of both halves are zeroed.
__global__ __launch_bounds__(256) void split_x3_kernel(const float* __restrict__ x, int64_t rows, int K, int ldx, half_t* __restrict__ out,
                                                       int ldo, int Kp, int swap) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kq = Kp >> 2;
  if (i >= rows * kq) return;
  const int64_t r = i / kq;
  const int c = (int)(i - r * kq) * 4;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = c + e < K ? x[(size_t)r * ldx + c + e] : 0.f;
    const half_t hv = (half_t)v;
    hi[e] = hv;
    lo[e] = (half_t)((v - (float)hv) * 2048.0f);
  }
  half_t* o = out + (size_t)r * ldo + c;
  *reinterpret_cast<h4*>(o + (swap ? Kp : 0)) = hi;
  *reinterpret_cast<h4*>(o + (swap ? 0 : Kp)) = lo;
}

void launch_split_x3(hipStream_t s, const float* x, int64_t rows, int K, int ldx, half_t* out, int ldo, int Kp, int swap) {
  if (rows <= 0) return;
  const int64_t n = rows * (Kp >> 2);
  hipLaunchKernelGGL(split_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, rows, K, ldx, out, ldo, Kp, swap);
  PF_HIP(hipGetLastError());
}

// im2col of the CIF conv on fp32 rows: out[(b,t), j*D + c] = H[b, t+j-l, c] (0 outside the utterance)
__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ Hm, int B, int T, int D, int l_order, int taps,
                                                         float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * T * taps * D;
  if (i >= total) return;
  const int c = (int)(i % D);
  int64_t r = i / D;
  const int j = (int)(r % taps);
  const int64_t row = r / taps;
  const int t = (int)(row % T), tt = t + j - l_order;
  out[i] = (tt >= 0 && tt < T) ? Hm[(row + (j - l_order)) * (int64_t)D + c] : 0.f;
}

void launch_im2col_f32(hipStream_t s, const float* Hm, int B, int T, int D, int l_order, int r_order, float* out) {
  const int taps = l_order + r_order + 1;
  const int64_t total = (int64_t)B * T * taps * D;
  if (total == 0) return;
  hipLaunchKernelGGL(im2col_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Hm, B, T, D, l_order, taps, out);
  PF_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void add_f32_kernel(float* __restrict__ x, const float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += y[i];
}
void launch_add_f32(hipStream_t s, float* x, const float* y, int64_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
  PF_HIP(hipGetLastError());
}

// x*sqrt(d_model) + PE (two roundings, as the Mul and Add nodes) -> fp32 rows (the fp32 mode keeps the embed stage apart)
__global__ __launch_bounds__(256) void posenc_f32_kernel(const float* __restrict__ x, const float* __restrict__ pe, int64_t rows, int T,
                                                         int F, float xscale, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const int64_t row = i / F;
  const int c = (int)(i - row * F);
  out[i] = add_rn(mul_rn(x[i], xscale), pe[(row % T) * (int64_t)F + c]);   // no contraction to an FMA: pad rows sit at |x| ~ 1.7e7
}

void launch_posenc_f32(hipStream_t s, const float* x, const float* pe, int B, int T, int F, float xscale, float* out) {
  const int64_t n = (int64_t)B * T * F;
  if (n == 0) return;
  hipLaunchKernelGGL(posenc_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, pe, (int64_t)B * T, T, F, xscale, out);
  PF_HIP(hipGetLastError());
}

// LSTM cell on fp32 gate pre-activations g[b, 0:4D] (PyTorch order i, f, g, o; already xg + h W_hh^T): c, h updated in
// place, h also written to hout[b * hout_bs + k] (the sequence output)
__global__ __launch_bounds__(256) void lstm_cell_f32_kernel(const float* __restrict__ g, int ldg, float* __restrict__ c, float* __restrict__ h,
                                                            float* __restrict__ hout, int64_t hout_bs, int B, int D) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * D) return;
  const int b = idx / D, k = idx - b * D;
  const float* gr = g + (size_t)b * ldg;
  auto sig = [](float x) { return 1.0f / (1.0f + expf(-x)); };
  const float ig = sig(gr[k]), fg = sig(gr[D + k]), gg = tanhf(gr[2 * D + k]), og = sig(gr[3 * D + k]);
  const float cn = add_rn(mul_rn(fg, c[idx]), mul_rn(ig, gg));
  const float hn = mul_rn(og, tanhf(cn));
  c[idx] = cn;
  h[idx] = hn;
  hout[(size_t)b * hout_bs + k] = hn;
}

void launch_lstm_cell_f32(hipStream_t s, const float* gates, int ldg, float* c, float* h, float* hout, int64_t hout_bs, int B, int D) {
  if (B * D == 0) return;
  hipLaunchKernelGGL(lstm_cell_f32_kernel, dim3((unsigned)cdiv((int64_t)B * D, 256)), dim3(256), 0, s, gates, ldg, c, h, hout, hout_bs, B, D);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
