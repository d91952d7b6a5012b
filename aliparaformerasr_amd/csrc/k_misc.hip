// k_misc.hip — HBM-/latency-bound side kernels: DFSMN memory blocks, CIF predictor tail,
// integrate-and-fire, last-index arg-max.
//
// Reference sites:
//   * arg-max: AliParaformerAsr/OfflineRecognizer.cs:139-152 — cur = x[cur] > x[k] ? cur : k,
//     so ties and NaN compares resolve to the LARGER index (quirk Q4).
//   * CIF: semantics of the sequential integrate-and-fire are stated in-tree by the streaming
//     path, AliParaformerAsr/OnlineRecognizer.cs:147-200; thresholds/tail from
//     AliParaformerAsr/Model/PredictorConfEntity.cs:13-17.  The offline graph itself is
//     external (FunASR CifPredictorV2 export executed by onnxruntime).
//   * FSMN: external ONNX graph (FunASR MultiHeadedAttentionSANM.forward_fsmn and
//     MultiHeadedAttentionSANMDecoder); kernel_size 11 from EncoderConfEntity.cs:23.
#include "kernels.h"
#include "exact.h"

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ encoder FSMN ----------
// f[b,t,c] = sum_j wT[j][c] * v[b,t+j-left,c] + v[b,t,c]   (zero outside the utterance)
// One thread = 8 channels x FS_ROWS consecutive frames with a register sliding window: each v
// row is loaded once per FS_ROWS outputs (+ the K-1 halo rows) instead of K times.  v is f16
// (the V slice of the QKV buffer), f is fp32.  HBM-bound: 2 B/elem in, 4 B/elem out.
#ifndef FS_ROWS
#define FS_ROWS 8
#endif
// FS_CH channels per thread (8 -> one 16-byte f16 load per row, 4 -> 8-byte loads but half the registers
// and twice the threads in flight; selected at build time, see tools/fsmn_rows.sh)
#ifndef FS_CH
#define FS_CH 4
#endif
#ifndef FS_ST
#define FS_ST 2      // cache policy of the fp32 result stores: 0 plain, 1 nt, 2 sc1 (read two kernels later; A/B tools/fsmn_rows.sh: -3 %)
#endif
template <int K>
__global__ __launch_bounds__(256) void fsmn_enc_kernel(const half_t* __restrict__ v, int ldv,
                                                       const float* __restrict__ wT, int B, int T, int D,
                                                       float* __restrict__ f) {
  constexpr int C = FS_CH;
  typedef _Float16 hC __attribute__((ext_vector_type(C)));
  const int cq = D / C;
  const int tb = (T + FS_ROWS - 1) / FS_ROWS;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * tb * cq;
  if (i >= total) return;
  const int c0 = (int)(i % cq) * C;
  const int64_t r = i / cq;
  const int t0 = (int)(r % tb) * FS_ROWS;
  const int b = (int)(r / tb);
  constexpr int left = (K - 1) / 2;
  const half_t* vb = v + (int64_t)b * T * ldv + c0;
  float w[K][C];
#pragma unroll
  for (int j = 0; j < K; ++j)
#pragma unroll
    for (int e = 0; e < C; e += 4) {
      const float4 w0 = *reinterpret_cast<const float4*>(wT + (int64_t)j * D + c0 + e);
      w[j][e] = w0.x; w[j][e + 1] = w0.y; w[j][e + 2] = w0.z; w[j][e + 3] = w0.w;
    }
  float acc[FS_ROWS][C];
#pragma unroll
  for (int q = 0; q < FS_ROWS; ++q)
#pragma unroll
    for (int e = 0; e < C; ++e) acc[q][e] = 0.f;
  // input row tt = t0 - left + s contributes to output row q = s - j (tap j) for 0 <= q < FS_ROWS
#pragma unroll
  for (int s = 0; s < FS_ROWS + K - 1; ++s) {
    const int tt = t0 - left + s;
    if (tt < 0 || tt >= T) continue;
    const hC x = *reinterpret_cast<const hC*>(vb + (int64_t)tt * ldv);
    float xf[C];
#pragma unroll
    for (int e = 0; e < C; ++e) xf[e] = (float)x[e];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int q = s - j;
      if (q >= 0 && q < FS_ROWS) {
#pragma unroll
        for (int e = 0; e < C; ++e) acc[q][e] += w[j][e] * xf[e];
      }
    }
    const int q0 = s - left;                      // identity term
    if (q0 >= 0 && q0 < FS_ROWS) {
#pragma unroll
      for (int e = 0; e < C; ++e) acc[q0][e] += xf[e];
    }
  }
#pragma unroll
  for (int q = 0; q < FS_ROWS; ++q) {
    if (t0 + q < T) {
      float4* o = reinterpret_cast<float4*>(f + ((int64_t)b * T + t0 + q) * D + c0);
#pragma unroll
      for (int e = 0; e < C; e += 4) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v4 = {acc[q][e], acc[q][e + 1], acc[q][e + 2], acc[q][e + 3]};
#if FS_ST == 2
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o + e / 4), "v"(v4) : "memory");
#elif FS_ST == 1
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o + e / 4), "v"(v4) : "memory");
#else
        *reinterpret_cast<f4v*>(o + e / 4) = v4;
#endif
      }
    }
  }
}

void launch_fsmn_enc(hipStream_t s, const half_t* v, int ldv, const float* wT, int B, int T, int D, int k,
                     float* f) {

  const int tb = (T + FS_ROWS - 1) / FS_ROWS;
  const int64_t total = (int64_t)B * tb * (D / FS_CH);
  if (total == 0) return;
  const dim3 grid((unsigned)((total + 255) / 256));
  switch (k) {
    case 11: hipLaunchKernelGGL(fsmn_enc_kernel<11>, grid, dim3(256), 0, s, v, ldv, wT, B, T, D, f); break;
    case 9: hipLaunchKernelGGL(fsmn_enc_kernel<9>, grid, dim3(256), 0, s, v, ldv, wT, B, T, D, f); break;
    case 7: hipLaunchKernelGGL(fsmn_enc_kernel<7>, grid, dim3(256), 0, s, v, ldv, wT, B, T, D, f); break;
    case 5: hipLaunchKernelGGL(fsmn_enc_kernel<5>, grid, dim3(256), 0, s, v, ldv, wT, B, T, D, f); break;
    case 3: hipLaunchKernelGGL(fsmn_enc_kernel<3>, grid, dim3(256), 0, s, v, ldv, wT, B, T, D, f); break;
    default: throw Error(PF_ERR_UNSUPPORTED, "fsmn: unsupported kernel size (3/5/7/9/11)");
  }
  PF_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ fp32 FSMN -------------
// y = (dwconv(v*m) + v*m) * m ; m from token_num (l < token_num[b]) or a float mask or none.
// accumulate != 0: out += y (decoder residual), else out = y.
// KT > 0: compile-time tap count — all K rows are requested before the first FMA (the runtime-K loop
// below serialises one L2 round trip per tap, which is what bounds the small decoder launches).
template <int KT>
__global__ __launch_bounds__(256) void fsmn_f32_kernel(const float* __restrict__ v, const float* __restrict__ wT,
                                                       const float* __restrict__ mask,
                                                       const int32_t* __restrict__ token_num, int B, int T,
                                                       int D, int Krt, int accumulate, float* __restrict__ out) {
  const int K = KT > 0 ? KT : Krt;
  const int cq = D >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * T * cq;
  if (i >= total) return;
  const int c4 = (int)(i % cq) * 4;
  const int64_t row = i / cq;
  const int t = (int)(row % T);
  const int b = (int)(row / T);
  const int left = (K - 1) / 2;
  auto m_at = [&](int tt) -> float {
    if (token_num) return tt < token_num[b] ? 1.f : 0.f;
    if (mask) return mask[(int64_t)b * T + tt];
    return 1.f;
  };
  const float mt = m_at(t);
  float4 acc;
  {
    const float4 x = *reinterpret_cast<const float4*>(v + row * (int64_t)D + c4);
    acc = make_float4(x.x * mt, x.y * mt, x.z * mt, x.w * mt);
  }
  if constexpr (KT > 0) {
    float4 xs[KT > 0 ? KT : 1], ws[KT > 0 ? KT : 1];
    float mm[KT > 0 ? KT : 1];
    bool ok[KT > 0 ? KT : 1];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int tt = t + j - left;
      ok[j] = tt >= 0 && tt < T;
      const int ttc = ok[j] ? tt : t;
      xs[j] = *reinterpret_cast<const float4*>(v + (row + (ttc - t)) * (int64_t)D + c4);
      ws[j] = *reinterpret_cast<const float4*>(wT + (int64_t)j * D + c4);
      mm[j] = m_at(ttc);
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      if (ok[j]) {
        acc.x += ws[j].x * (xs[j].x * mm[j]); acc.y += ws[j].y * (xs[j].y * mm[j]);
        acc.z += ws[j].z * (xs[j].z * mm[j]); acc.w += ws[j].w * (xs[j].w * mm[j]);
      }
    }
  } else {
    for (int j = 0; j < K; ++j) {
      const int tt = t + j - left;
      if (tt < 0 || tt >= T) continue;
      const float mm = m_at(tt);
      const float4 x = *reinterpret_cast<const float4*>(v + (row + (j - left)) * (int64_t)D + c4);
      const float4 w = *reinterpret_cast<const float4*>(wT + (int64_t)j * D + c4);
      acc.x += w.x * (x.x * mm); acc.y += w.y * (x.y * mm);
      acc.z += w.z * (x.z * mm); acc.w += w.w * (x.w * mm);
    }
  }
  acc.x *= mt; acc.y *= mt; acc.z *= mt; acc.w *= mt;
  float4* o = reinterpret_cast<float4*>(out + row * (int64_t)D + c4);
  if (accumulate) {
    const float4 p = *o;
    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
  }
  *o = acc;
}

// decoder FSMN with the encoder kernel's register sliding window: one thread = 4 channels x FS_ROWS
// consecutive positions; x[b,l,:] += (sum_j w_j * tn[l+j-left]*m + tn[l]*m) * m_l, m = (l < token_num[b]).
// (The one-position-per-thread form above re-reads every row K times through L1: 21 us for 11 MB.)
template <int K>
__global__ __launch_bounds__(256) void fsmn_dec_kernel(const float* __restrict__ tn, const float* __restrict__ wT,
                                                       const int32_t* __restrict__ token_num, int B, int L, int D,
                                                       float* __restrict__ x) {
  const int cq = D >> 2;
  const int tb = (L + FS_ROWS - 1) / FS_ROWS;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * tb * cq;
  if (i >= total) return;
  const int c4 = (int)(i % cq) * 4;
  const int64_t r = i / cq;
  const int t0 = (int)(r % tb) * FS_ROWS;
  const int b = (int)(r / tb);
  constexpr int left = (K - 1) / 2;
  const int nvalid = token_num[b];
  const float* vb = tn + (int64_t)b * L * D + c4;
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = *reinterpret_cast<const float4*>(wT + (int64_t)j * D + c4);
  float4 acc[FS_ROWS];
#pragma unroll
  for (int q = 0; q < FS_ROWS; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < FS_ROWS + K - 1; ++s) {
    const int tt = t0 - left + s;
    if (tt < 0 || tt >= L || tt >= nvalid) continue;      // masked rows contribute exact zeros
    const float4 xv = *reinterpret_cast<const float4*>(vb + (int64_t)tt * D);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int q = s - j;
      if (q >= 0 && q < FS_ROWS) {
        acc[q].x += w[j].x * xv.x; acc[q].y += w[j].y * xv.y; acc[q].z += w[j].z * xv.z; acc[q].w += w[j].w * xv.w;
      }
    }
    const int q0 = s - left;
    if (q0 >= 0 && q0 < FS_ROWS) { acc[q0].x += xv.x; acc[q0].y += xv.y; acc[q0].z += xv.z; acc[q0].w += xv.w; }
  }
#pragma unroll
  for (int q = 0; q < FS_ROWS; ++q) {
    const int l = t0 + q;
    if (l < L && l < nvalid) {                              // masked outputs add zero: x unchanged
      float4* o = reinterpret_cast<float4*>(x + ((int64_t)b * L + l) * D + c4);
      const float4 pv = *o;
      *o = make_float4(pv.x + acc[q].x, pv.y + acc[q].y, pv.z + acc[q].z, pv.w + acc[q].w);
    }
  }
}

void launch_fsmn_dec(hipStream_t s, const float* tn, const float* wT, const int32_t* token_num, int B, int L,
                     int D, int k, float* x) {
  const int64_t total = (int64_t)B * L * (D / 4);
  if (total == 0) return;
  const int64_t tot = (int64_t)B * ((L + FS_ROWS - 1) / FS_ROWS) * (D / 4);
  if (k == 11) {
    hipLaunchKernelGGL(fsmn_dec_kernel<11>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, tn, wT, token_num,
                       B, L, D, x);
  } else if (k == 21) {                               // SeACo bias decoder
    hipLaunchKernelGGL(fsmn_dec_kernel<21>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, tn, wT, token_num,
                       B, L, D, x);
  } else
    hipLaunchKernelGGL(fsmn_f32_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tn, wT,
                       (const float*)nullptr, token_num, B, L, D, k, 1, x);
  PF_HIP(hipGetLastError());
}

// One thread per (utterance, channel): walks the K-1 cached columns and the L new positions.  Tiny by construction
// (a streaming chunk yields a handful of tokens), so clarity over bandwidth.
__global__ __launch_bounds__(256) void fsmn_dec_stream_kernel(const float* __restrict__ tn, const float* __restrict__ wT,
                                                              const int32_t* __restrict__ len, const float* __restrict__ cache_in,
                                                              int B, int L, int D, int K, float* __restrict__ x,
                                                              float* __restrict__ cache_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * D) return;
  const int b = (int)(i / D), c = (int)(i - (int64_t)b * D);
  const int CW = K - 1, nv = len[b];
  auto xc = [&](int p) -> float {                       // concatenation [cache | masked new positions]
    if (p < CW) return cache_in[((int64_t)b * D + c) * CW + p];
    const int l = p - CW;
    return l < nv ? tn[((int64_t)b * L + l) * D + c] : 0.f;
  };
  for (int l = 0; l < L; ++l) {
    if (l < nv) {
      float acc = 0.f;
      for (int j = 0; j < K; ++j) acc += wT[(int64_t)j * D + c] * xc(l + j);
      acc += tn[((int64_t)b * L + l) * D + c];
      x[((int64_t)b * L + l) * D + c] += acc;
    }
  }
  for (int q = 0; q < CW; ++q) cache_out[((int64_t)b * D + c) * CW + q] = xc(L + q);
}

void launch_fsmn_dec_stream(hipStream_t s, const float* tn, const float* wT, const int32_t* len, const float* cache_in,
                            int B, int L, int D, int k, float* x, float* cache_out) {
  const int64_t total = (int64_t)B * D;
  if (total == 0) return;
  hipLaunchKernelGGL(fsmn_dec_stream_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tn, wT, len, cache_in,
                     B, L, D, k, x, cache_out);
  PF_HIP(hipGetLastError());
}

// fp32 encoder FSMN without a mask (math_mode 1 / 3), register window: one thread = 4 channels x FS_ROWS consecutive frames; the
// FS_ROWS + K - 1 input rows it needs are requested up front, then every output is formed in the order of fsmn_f32_kernel
// (identity term first, taps 0 .. K-1) — the same roundings, a third of the time (that form re-reads every row K times
// through L1 with one L2 round trip per tap: 38 us for 2 x 32 MB at the benchmark shape).
template <int K>
__global__ __launch_bounds__(256) void fsmn_f32_win_kernel(const float* __restrict__ v, int ldv, const float* __restrict__ wT, int B, int T, int D,
                                                           float* __restrict__ out) {
  const int cq = D >> 2;
  const int tb = (T + FS_ROWS - 1) / FS_ROWS;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * tb * cq) return;
  const int c4 = (int)(i % cq) * 4;
  const int64_t r = i / cq;
  const int t0 = (int)(r % tb) * FS_ROWS;
  const int b = (int)(r / tb);
  constexpr int left = (K - 1) / 2, NR = FS_ROWS + K - 1;
  const float* vb = v + (int64_t)b * T * ldv + c4;
  float4 x[NR];
#pragma unroll
  for (int s = 0; s < NR; ++s) {
    int tt = t0 - left + s;
    tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);               // clamped reads; rows outside [0, T) are skipped below
    x[s] = *reinterpret_cast<const float4*>(vb + (int64_t)tt * ldv);
  }
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = *reinterpret_cast<const float4*>(wT + (int64_t)j * D + c4);
#pragma unroll
  for (int q = 0; q < FS_ROWS; ++q) {
    const int t = t0 + q;
    if (t >= T) break;
    float4 acc = x[q + left];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int tt = t + j - left;
      if (tt >= 0 && tt < T) {
        acc.x += w[j].x * x[q + j].x; acc.y += w[j].y * x[q + j].y; acc.z += w[j].z * x[q + j].z; acc.w += w[j].w * x[q + j].w;
      }
    }
    *reinterpret_cast<float4*>(out + ((int64_t)b * T + t) * D + c4) = acc;
  }
}

// v rows with stride ldv (the V third of a fused [M, 3 D] Q | K | V result), no mask, k = 11, v and y distinct
void launch_fsmn_f32_ld(hipStream_t s, const float* v, int ldv, const float* wT, int B, int T, int D, int k, float* y) {
  PF_CHECK(k == 11 && ldv % 4 == 0 && D % 4 == 0, PF_ERR_UNSUPPORTED, "fsmn_f32_ld: kernel size 11, 16-byte rows");
  const int64_t tot = (int64_t)B * ((T + FS_ROWS - 1) / FS_ROWS) * (D / 4);
  if (tot == 0) return;
  hipLaunchKernelGGL(fsmn_f32_win_kernel<11>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, v, ldv, wT, B, T, D, y);
  PF_HIP(hipGetLastError());
}

void launch_fsmn_f32(hipStream_t s, const float* v, const float* wT, const float* mask, int B, int T, int D,
                     int k, float* y) {
  const int64_t total = (int64_t)B * T * (D / 4);
  if (total == 0) return;
  if (!mask && k == 11 && v != y) { launch_fsmn_f32_ld(s, v, D, wT, B, T, D, k, y); return; }
  hipLaunchKernelGGL(fsmn_f32_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, v, wT, mask,
                     (const int32_t*)nullptr, B, T, D, k, 0, y);
  PF_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ CIF predictor ---------
// im2col for the k = l+r+1 conv: out[(b,t), j*D + c] = H[b, t+j-l, c] (0 outside)
__global__ __launch_bounds__(256) void cif_im2col_kernel(const half_t* __restrict__ H, int B, int T, int D,
                                                         int l_order, int taps, half_t* __restrict__ out) {
  const int cq = D >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * T * taps * cq;
  if (i >= total) return;
  const int c8 = (int)(i % cq) * 8;
  int64_t r = i / cq;
  const int j = (int)(r % taps);
  const int64_t row = r / taps;
  const int t = (int)(row % T);
  const int tt = t + j - l_order;
  h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
  if (tt >= 0 && tt < T) x = *reinterpret_cast<const h8*>(H + (row + (j - l_order)) * (int64_t)D + c8);
  *reinterpret_cast<h8*>(out + row * (int64_t)(taps * D) + (int64_t)j * D + c8) = x;
}

void launch_cif_im2col(hipStream_t s, const half_t* H, int B, int T, int D, int l_order, int r_order,
                       half_t* out) {
  const int taps = l_order + r_order + 1;
  const int64_t total = (int64_t)B * T * taps * (D / 8);
  if (total == 0) return;
  hipLaunchKernelGGL(cif_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, H, B, T, D,
                     l_order, taps, out);
  PF_HIP(hipGetLastError());
}

// one wavefront per (b,t): alpha = relu(sigmoid(y . w + b0) * smooth - noise); alphas[b,T] = tail
__global__ __launch_bounds__(256) void cif_alpha_kernel(const float* __restrict__ y, int B, int T, int D,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b0, float smooth, float noise,
                                                        float tail, float* __restrict__ alphas) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * T) return;
  const float4* yr = reinterpret_cast<const float4*>(y + row * (int64_t)D);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float acc = 0.f;
  for (int qd = lane; qd < (D >> 2); qd += 64) {
    const float4 a = yr[qd], c = w4[qd];
    acc += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) {
    const float z = acc + b0[0];
    float a = 1.0f / (1.0f + expf(-z));
    a = a * smooth - noise;
    a = a > 0.f ? a : 0.f;
    const int b = (int)(row / T), t = (int)(row % T);
    alphas[(int64_t)b * (T + 1) + t] = a;
    if (t == T - 1) alphas[(int64_t)b * (T + 1) + T] = tail;
  }
}

void launch_cif_alpha(hipStream_t s, const float* y, int B, int T, int D, const float* w, const float* b0,
                      float smooth, float noise, float tail, float* alphas) {
  const int64_t rows = (int64_t)B * T;
  if (rows == 0) return;
  hipLaunchKernelGGL(cif_alpha_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, y, B, T, D, w, b0,
                     smooth, noise, tail, alphas);
  PF_HIP(hipGetLastError());
}

// Sequential integrate-and-fire, one wavefront per utterance (lane 0 walks the T1 frames; the
// recurrence is a 3-op fp32 dependency chain, the alphas are prefetched to LDS by all lanes).
// Arithmetic order is exactly the sequential definition (integrate += alpha; fire when >= thr;
// integrate -= 1; cur = 1 - integrate_before; remainder = alpha - cur).
__global__ __launch_bounds__(64) void cif_scan_kernel(const float* __restrict__ alphas, int B, int T1,
                                                      float threshold, CifPlan plan) {
  extern __shared__ float sa[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* a = alphas + (int64_t)b * T1;
  for (int t = lane; t < T1; t += 64) sa[t] = a[t];
  __syncthreads();
  if (lane != 0) return;
  float integrate = 0.f, sum = 0.f;
  int count = 0;
  int32_t* ff = plan.fire_frame + (int64_t)b * T1;
  float* wc = plan.w_cur + (int64_t)b * T1;
  float* wr = plan.w_rem + (int64_t)b * T1;
  for (int t = 0; t < T1; ++t) {
    const float alpha = sa[t];
    sum = add_rn(sum, alpha);
    const float completion = sub_rn(1.0f, integrate);
    integrate = add_rn(integrate, alpha);
    if (integrate >= threshold) {
      integrate = sub_rn(integrate, 1.0f);
      wc[t] = completion;
      wr[t] = sub_rn(alpha, completion);
      ff[count++] = t;
    } else {
      wc[t] = alpha;
      wr[t] = 0.f;
    }
  }
  plan.fire_count[b] = count;
  plan.token_num[b] = (int32_t)floorf(sum);
  atomicMax(plan.max_count, count);
}

void launch_cif_scan(hipStream_t s, const float* alphas, int B, int T1, float threshold, CifPlan plan) {
  if (B == 0) return;
  PF_HIP(hipMemsetAsync(plan.max_count, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(cif_scan_kernel, dim3(B), dim3(64), sizeof(float) * T1, s, alphas, B, T1, threshold, plan);
  PF_HIP(hipGetLastError());
}

// ---- the prefix-sum formulation (FunASR cif_v1_export; container key cif_variant = "cumsum") ----------------
// prefix = float32(cumsum_float64(alpha)); fire at t <=> floor(prefix[t]) > floor(prefix[t-1]);
// remain[t] = (1 + (prefix[t] - floor(prefix[t]))) - 1.
// A WAVE per utterance: lane i owns the contiguous chunk [i c, (i + 1) c) of the T + 1 weights, sums it in double, the
// 64 chunk totals go through a wave prefix scan, and every lane rebuilds its own prefixes from its offset.  Double
// additions of fp32 weights are exact here except in contrived cases (a sum needs more than 53 bits only when weights
// 2^30 apart in magnitude meet), and a re-associated sum equals the sequential ONNX CumSum bit for bit whenever every
// addition was exact — which each addition checks (Knuth's TwoSum); an utterance with ANY inexact addition is redone
// sequentially by lane 0, so the result is the sequential one by construction.  token_num = floor(fp32 sequential sum)
// is an order-dependent fp32 chain and stays one (lane 0, ~T dependent adds).
__device__ __forceinline__ double two_sum(double a, double b, bool& inexact) {
#pragma clang fp contract(off)
  const double s = a + b;
  const double bb = s - a;
  const double err = (a - (s - bb)) + (b - bb);
  inexact |= err != 0.0;
  return s;
}

__global__ __launch_bounds__(64) void cif_scan_cumsum_kernel(const float* __restrict__ alphas, int B, int T1, CifPlan plan) {
  extern __shared__ float sa[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* a = alphas + (int64_t)b * T1;
  for (int t = lane; t < T1; t += 64) sa[t] = a[t];
  __syncthreads();
  int32_t* ff = plan.fire_frame + (int64_t)b * T1;
  float* wr = plan.w_rem + (int64_t)b * T1;
  const int c = (T1 + 63) / 64;                         // chunk length
  const int t0 = min(lane * c, T1), t1 = min(t0 + c, T1);
  bool inexact = false;
  double tot = 0.0;
  for (int t = t0; t < t1; ++t) tot = two_sum(tot, (double)sa[t], inexact);
  // inclusive wave scan of the chunk totals (Hillis-Steele over 64 lanes), then exclusive offset
  double inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(inc, o, 64);
    if (lane >= o) inc = two_sum(up, inc, inexact);
  }
  double off = __shfl_up(inc, 1, 64);
  if (lane == 0) off = 0.0;
  // floor of the prefix just before this chunk (the last element of the previous non-empty chunk)
  float prev_floor = floorf((float)off);
  if (t0 == 0) prev_floor = 0.f;
  int nfire = 0;
  double run = off;
  for (int t = t0; t < t1; ++t) {
    run = two_sum(run, (double)sa[t], inexact);
    const float fl = floorf((float)run);
    nfire += sub_rn(fl, prev_floor) > 0.f ? 1 : 0;
    prev_floor = fl;
  }
  const bool any_inexact = __any(inexact);
  if (!any_inexact) {
    int pos = nfire;                                    // exclusive scan of the fire counts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(pos, o, 64);
      if (lane >= o) pos += up;
    }
    const int total = __shfl(pos, 63, 64);
    pos -= nfire;
    run = off;
    prev_floor = t0 == 0 ? 0.f : floorf((float)off);
    for (int t = t0; t < t1; ++t) {
      run += (double)sa[t];                             // exact (checked above): same value as in the first walk
      const float prefix = (float)run;
      const float fl = floorf(prefix);
      if (sub_rn(fl, prev_floor) > 0.f) {
        const float fires = add_rn(1.0f, sub_rn(prefix, fl));
        wr[t] = sub_rn(fires, floorf(fires));
        ff[pos++] = t;
      } else {
        wr[t] = 0.f;
      }
      prev_floor = fl;
    }
    if (lane == 0) {
      float sum = 0.f;
      for (int t = 0; t < T1; ++t) sum = add_rn(sum, sa[t]);
      plan.fire_count[b] = total;
      plan.token_num[b] = (int32_t)floorf(sum);
      atomicMax(plan.max_count, total);
    }
    return;
  }
  if (lane != 0) return;                                // some addition was inexact: the sequential definition, verbatim
  double dsum = 0.0;
  float sum = 0.f;
  prev_floor = 0.f;
  int count = 0;
  for (int t = 0; t < T1; ++t) {
    const float alpha = sa[t];
    sum = add_rn(sum, alpha);
    dsum += (double)alpha;
    const float prefix = (float)dsum;
    const float fl = floorf(prefix);
    if (sub_rn(fl, prev_floor) > 0.f) {
      const float fires = add_rn(1.0f, sub_rn(prefix, fl));
      wr[t] = sub_rn(fires, floorf(fires));
      ff[count++] = t;
    } else {
      wr[t] = 0.f;
    }
    prev_floor = fl;
  }
  plan.fire_count[b] = count;
  plan.token_num[b] = (int32_t)floorf(sum);
  atomicMax(plan.max_count, count);
}

void launch_cif_scan_cumsum(hipStream_t s, const float* alphas, int B, int T1, CifPlan plan) {
  if (B == 0) return;
  PF_HIP(hipMemsetAsync(plan.max_count, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(cif_scan_cumsum_kernel, dim3(B), dim3(64), sizeof(float) * T1, s, alphas, B, T1, plan);
  PF_HIP(hipGetLastError());
}

// E[l] = ((psh[f_l] - psh[f_{l-1}]) + remain[f_{l-1}] H[f_{l-1}]) - remain[f_l] H[f_l], psh = sequential fp32 running sum of
// alpha[t] H[t] (ONNX CumSum).  One thread per (utterance, 4 channels) walks the T+1 frames; rows l >= fire_count are zero.
__global__ __launch_bounds__(128) void cif_gather_cumsum_kernel(const float* __restrict__ H, const float* __restrict__ alphas,
                                                                int B, int T, int D, int T1, CifPlan plan, int L,
                                                                float* __restrict__ E) {
  const int b = blockIdx.x;
  const int cnt = plan.fire_count[b];
  const int32_t* ff = plan.fire_frame + (int64_t)b * T1;
  const float* wr = plan.w_rem + (int64_t)b * T1;
  const float* al = alphas + (int64_t)b * T1;
  for (int c4 = threadIdx.x * 4; c4 < D; c4 += blockDim.x * 4) {
    float4 psh = make_float4(0.f, 0.f, 0.f, 0.f), last = psh, lrem = psh;
    int l = 0;
    int next = cnt > 0 ? ff[0] : T1;
    for (int t = 0; t < T1; ++t) {
      float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < T) h = *reinterpret_cast<const float4*>(H + ((int64_t)b * T + t) * D + c4);
      const float a = al[t];
      psh.x = add_rn(psh.x, mul_rn(a, h.x)); psh.y = add_rn(psh.y, mul_rn(a, h.y));
      psh.z = add_rn(psh.z, mul_rn(a, h.z)); psh.w = add_rn(psh.w, mul_rn(a, h.w));
      if (t == next) {
        const float r = wr[t];
        const float4 rh = make_float4(mul_rn(r, h.x), mul_rn(r, h.y), mul_rn(r, h.z), mul_rn(r, h.w));
        float4 o;
        o.x = sub_rn(add_rn(sub_rn(psh.x, last.x), lrem.x), rh.x); o.y = sub_rn(add_rn(sub_rn(psh.y, last.y), lrem.y), rh.y);
        o.z = sub_rn(add_rn(sub_rn(psh.z, last.z), lrem.z), rh.z); o.w = sub_rn(add_rn(sub_rn(psh.w, last.w), lrem.w), rh.w);
        if (l < L) *reinterpret_cast<float4*>(E + ((int64_t)b * L + l) * D + c4) = o;
        last = psh; lrem = rh;
        ++l;
        next = l < cnt ? ff[l] : T1;
      }
    }
    for (; l < L; ++l) *reinterpret_cast<float4*>(E + ((int64_t)b * L + l) * D + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

void launch_cif_gather_cumsum(hipStream_t s, const float* H, const float* alphas, int B, int T, int D, int T1, CifPlan plan,
                              int L, float* E) {
  if (B == 0 || L == 0) return;
  hipLaunchKernelGGL(cif_gather_cumsum_kernel, dim3(B), dim3(128), 0, s, H, alphas, B, T, D, T1, plan, L, E);
  PF_HIP(hipGetLastError());
}

// E[b,l,:] = w_rem[s]*H[s] + sum_{t in (s,e]} w_cur[t]*H[t], s = fire_frame[l-1], e = fire_frame[l]
// (l = 0: from frame 0, no carried remainder).  Frames t >= T are the zero tail frame.
// Products and sums are separately rounded in ascending t, as the sequential loop does.
__global__ __launch_bounds__(128) void cif_gather_kernel(const float* __restrict__ H, int B, int T, int D, int T1,
                                                         CifPlan plan, int L, float* __restrict__ E) {
  const int bl = blockIdx.x;
  const int b = bl / L, l = bl - b * L;
  const int cnt = plan.fire_count[b];
  float* out = E + (int64_t)bl * D;
  const int32_t* ff = plan.fire_frame + (int64_t)b * T1;
  const float* wc = plan.w_cur + (int64_t)b * T1;
  const float* wr = plan.w_rem + (int64_t)b * T1;
  for (int c4 = threadIdx.x * 4; c4 < D; c4 += blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l < cnt) {
      const int e = ff[l];
      int t0 = 0;
      if (l > 0) {
        const int sfr = ff[l - 1];
        t0 = sfr + 1;
        if (sfr < T) {
          const float w = wr[sfr];
          const float4 h = *reinterpret_cast<const float4*>(H + ((int64_t)b * T + sfr) * D + c4);
          acc = make_float4(mul_rn(w, h.x), mul_rn(w, h.y), mul_rn(w, h.z), mul_rn(w, h.w));
        }
      }
      for (int t = t0; t <= e && t < T; ++t) {
        const float w = wc[t];
        const float4 h = *reinterpret_cast<const float4*>(H + ((int64_t)b * T + t) * D + c4);
        acc.x = add_rn(acc.x, mul_rn(w, h.x));
        acc.y = add_rn(acc.y, mul_rn(w, h.y));
        acc.z = add_rn(acc.z, mul_rn(w, h.z));
        acc.w = add_rn(acc.w, mul_rn(w, h.w));
      }
    }
    *reinterpret_cast<float4*>(out + c4) = acc;
  }
}

void launch_cif_gather(hipStream_t s, const float* H, int B, int T, int D, int T1, CifPlan plan, int L,
                       float* E) {
  if (B == 0 || L == 0) return;
  hipLaunchKernelGGL(cif_gather_kernel, dim3(B * L), dim3(128), 0, s, H, B, T, D, T1, plan, L, E);
  PF_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ SeACo -----------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                           int rows, int D, int vocab, float* __restrict__ out32,
                                                           half_t* __restrict__ out16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int dq = D >> 2;
  if (i >= (int64_t)rows * dq) return;
  const int r = (int)(i / dq), c4 = (int)(i - (int64_t)r * dq) * 4;
  int id = ids[r];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4 v = *reinterpret_cast<const float4*>(table + (int64_t)id * D + c4);
  *reinterpret_cast<float4*>(out32 + (int64_t)r * D + c4) = v;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  if (out16) *reinterpret_cast<h4*>(out16 + (int64_t)r * D + c4) = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
}

void launch_embed_gather(hipStream_t s, const float* table, const int32_t* ids, int rows, int D, int vocab,
                         float* out32, half_t* out16) {
  if (rows == 0) return;
  const int64_t total = (int64_t)rows * (D / 4);
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, table, ids, rows, D,
                     vocab, out32, out16);
  PF_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void add_to_f16_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t n4, half_t* __restrict__ out16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  reinterpret_cast<h4*>(out16)[i] = h4{(half_t)(x.x + y.x), (half_t)(x.y + y.y), (half_t)(x.z + y.z), (half_t)(x.w + y.w)};
}

void launch_add_to_f16(hipStream_t s, const float* a, const float* b, int64_t rows, int D, half_t* out16) {
  const int64_t n4 = rows * (D / 4);
  if (n4 == 0) return;
  hipLaunchKernelGGL(add_to_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a, b, n4, out16);
  PF_HIP(hipGetLastError());
}

// NO-BIAS decision + merge (_merge_res of the SeACo export with seaco_weight = 1): the graph's ArgMax keeps
// the FIRST maximal index, so "argmax == nobias" <=> x[nobias] > every earlier entry and >= every later one.
__global__ __launch_bounds__(256) void seaco_merge_kernel(const float* __restrict__ dha, int ld_dha,
                                                          const int64_t* __restrict__ dha_ids, int V, int nobias,
                                                          int copy_logits, float* __restrict__ logits, int ld_logits,
                                                          int64_t* __restrict__ ids) {
  __shared__ int s_bad[4];
  const int64_t row = blockIdx.x;
  const float* xr = dha + row * (int64_t)ld_dha;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int bad = 0;
  if (nobias >= 0 && nobias < V) {
    const float ref = xr[nobias];
    for (int k = tid; k < V; k += 256) {
      const float v = xr[k];
      if (k < nobias ? (v >= ref) : (k > nobias && v > ref)) bad = 1;
    }
    if (ref != ref) bad = 1;
  } else {
    bad = 1;
  }
  bad = __any(bad) ? 1 : 0;
  if (lane == 0) s_bad[wv] = bad;
  __syncthreads();
  const int replace = s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3];
  if (!replace) return;
  if (tid == 0) ids[row] = dha_ids[row];
  if (copy_logits) {
    float* lr = logits + row * (int64_t)ld_logits;
    for (int k = tid; k < V; k += 256) lr[k] = xr[k];
  }
}

void launch_seaco_merge(hipStream_t s, const float* dha, int ld_dha, const int64_t* dha_ids, int64_t rows, int V,
                        int nobias, int copy_logits, float* logits, int ld_logits, int64_t* ids) {
  if (rows == 0) return;
  hipLaunchKernelGGL(seaco_merge_kernel, dim3((unsigned)rows), dim3(256), 0, s, dha, ld_dha, dha_ids, V, nobias,
                     copy_logits, logits, ld_logits, ids);
  PF_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ arg-max ---------------
// The reference scans the tensor the graph RETURNS, i.e. the log-probs (OfflineRecognizer.cs:139-152 reads
// out[0]; the FunASR export ends in LogSoftmax), not the raw logits: x - lse rounds at a coarser ulp than x
// when |lse| > |x|, two distinct logits can become EQUAL log-probs, and the loop then keeps the LARGER index
// (quirk Q4).  So the pipeline forms y = (x - max) - log(sum exp(x - max)) — the two-step form of onnxruntime's
// CPU LogSoftmax (MLAS: (Input + NegativeMaximum) - Logarithm) — and runs the reference loop on y, whether or
// not the caller wants the log-probs stored.
struct VI { float v; int i; };
// combine of two partial scans over disjoint index sets: larger value, then larger index
__device__ __forceinline__ VI vi_later(VI a, VI b) {
  if (a.v > b.v) return a;
  if (b.v > a.v) return b;
  return a.i > b.i ? a : b;
}

// MODE 0: scan the values as given (stand-alone op).  MODE 1: scan the log-probs, do not store them.
// MODE 2: scan the log-probs and store them in place.  NV > 0: the row (V <= 256 * NV) is held in registers
// (one global read); NV == 0: the row is re-read (it stays in this CU's L1/L2).
template <int MODE, int NV>
__global__ __launch_bounds__(256) void argmax_kernel(float* __restrict__ x, int64_t rows, int V, int ldx,
                                                     int64_t* __restrict__ ids) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ int s_nan[4];
  __shared__ float s_f[4];
  const int64_t row = blockIdx.x;
  float* xr = x + row * (int64_t)ldx;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float reg[NV > 0 ? NV : 1];
  if constexpr (NV > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = tid + 256 * j;
      reg[j] = k < V ? xr[k] : 0.f;
    }
  }
  auto raw = [&](int j, int k) __attribute__((always_inline)) -> float {
    if constexpr (NV > 0) return reg[j];
    else return xr[k];
  };
  float mx = 0.f, lg = 0.f;
  if constexpr (MODE != 0) {
    // max (NaN ignored by fmaxf; it resurfaces through the sum), then sum of exp(x - max)
    float m = -INFINITY;
    if constexpr (NV > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) if (tid + 256 * j < V) m = fmaxf(m, reg[j]);
    } else {
      for (int k = tid; k < V; k += 256) m = fmaxf(m, xr[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) s_f[wv] = m;
    __syncthreads();
    mx = fmaxf(fmaxf(s_f[0], s_f[1]), fmaxf(s_f[2], s_f[3]));
    __syncthreads();
    float sum = 0.f;
    if constexpr (NV > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) if (tid + 256 * j < V) sum += expf(reg[j] - mx);
    } else {
      for (int k = tid; k < V; k += 256) sum += expf(xr[k] - mx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_f[wv] = sum;
    __syncthreads();
    lg = logf((s_f[0] + s_f[1]) + (s_f[2] + s_f[3]));
  }
  auto val = [&](int j, int k) __attribute__((always_inline)) -> float {
    const float v = raw(j, k);
    if constexpr (MODE != 0) return sub_rn(sub_rn(v, mx), lg);
    else return v;
  };
  // pass 1: last NaN position (the reference scan restarts after every NaN) + plain last-index arg-max
  int last_nan = -1;
  VI best = {-INFINITY, -1};
  auto visit = [&](int j, int k) __attribute__((always_inline)) {
    const float v = val(j, k);
    if constexpr (MODE == 2) xr[k] = v;
    if (v != v) last_nan = k;
    else if (v >= best.v) { best.v = v; best.i = k; }   // k ascending per thread
  };
  if constexpr (NV > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) if (tid + 256 * j < V) visit(j, tid + 256 * j);
  } else {
    for (int k = tid, j = 0; k < V; k += 256, ++j) visit(j, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int on = __shfl_xor(last_nan, o, 64);
    last_nan = on > last_nan ? on : last_nan;
    VI ob; ob.v = __shfl_xor(best.v, o, 64); ob.i = __shfl_xor(best.i, o, 64);
    best = vi_later(best, ob);
  }
  if (lane == 0) { s_v[wv] = best.v; s_i[wv] = best.i; s_nan[wv] = last_nan; }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    VI o = {s_v[w], s_i[w]};
    best = w == 0 ? o : vi_later(best, o);
    last_nan = w == 0 ? s_nan[w] : (s_nan[w] > last_nan ? s_nan[w] : last_nan);
  }
  int result = best.i;
  if (last_nan >= 0) {                       // rare path: redo over the suffix after the last NaN
    __syncthreads();
    VI b2 = {-INFINITY, -1};
    if constexpr (NV > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int k = tid + 256 * j;
        if (k < V && k > last_nan) { const float v = val(j, k); if (v >= b2.v) { b2.v = v; b2.i = k; } }
      }
    } else {
      for (int k = tid; k < V; k += 256)
        if (k > last_nan) {                  // MODE 2 already stored y over the row (own k's only: visible)
          const float v = MODE == 2 ? xr[k] : val(0, k);
          if (v >= b2.v) { b2.v = v; b2.i = k; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      VI ob; ob.v = __shfl_xor(b2.v, o, 64); ob.i = __shfl_xor(b2.i, o, 64);
      b2 = vi_later(b2, ob);
    }
    if (lane == 0) { s_v[wv] = b2.v; s_i[wv] = b2.i; }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
      VI o = {s_v[w], s_i[w]};
      b2 = w == 0 ? o : vi_later(b2, o);
    }
    result = (last_nan == V - 1 || b2.i < 0) ? last_nan : b2.i;
    // all -inf suffix: every compare "x[cur] > x[k]" is false -> last index
    if (last_nan < V - 1 && b2.i < 0) result = V - 1;
  } else if (result < 0) {
    result = V - 1;                          // row of -inf only
  }
  if (tid == 0) ids[row] = result;
}

void launch_argmax(hipStream_t s, float* x, int64_t rows, int V, int ldx, int mode, int64_t* ids) {
  if (rows == 0) return;
  const dim3 g((unsigned)rows), b(256);
  const bool small = V <= 256 * 36;          // paraformer's 8404-entry vocabulary: 33 values per thread
  const bool mid = V <= 256 * 100;           // SenseVoice's 25 055: 98 values per thread, still ONE read of the row (the
                                             // re-reading form took 0.74 ms for 10 880 rows = 1.5 TB/s)
#define PF_AM(MODE)                                                                                    \
  do {                                                                                                 \
    if (small) hipLaunchKernelGGL((argmax_kernel<MODE, 36>), g, b, 0, s, x, rows, V, ldx, ids);      \
    else if (mid) hipLaunchKernelGGL((argmax_kernel<MODE, 100>), g, b, 0, s, x, rows, V, ldx, ids);  \
    else hipLaunchKernelGGL((argmax_kernel<MODE, 0>), g, b, 0, s, x, rows, V, ldx, ids);             \
  } while (0)
  if (mode == 0) PF_AM(0);
  else if (mode == 1) PF_AM(1);
  else PF_AM(2);
#undef PF_AM
  PF_HIP(hipGetLastError());
}

// ---- hardware-queue probe (round 6, Engine::own_hardware_queue): a single wave that spins for ~`cycles` shader clocks, and an empty kernel
__global__ void spin_kernel(unsigned long long cycles) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < (1 << 22); ++i) {
    if (__builtin_readcyclecounter() - t0 >= cycles) break;
    __builtin_amdgcn_s_sleep(8);
  }
}
__global__ void nop_kernel() {}
void launch_spin(hipStream_t s, unsigned long long cycles) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cycles); }
void launch_nop(hipStream_t s) { hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, s); }

// The decoder length and the per-utterance counts of the CIF plan, written straight into PINNED HOST memory (dst is the device alias
// of a hipHostMalloc'ed buffer: [0] = max_count, [1 .. B] = fire_count, [1 + B .. 2 B] = token_num).  The engine waits for an event
// behind this kernel instead of queueing three device-to-host copies — copies are kernels of their own on this part and could not
// start while the K / V projection GEMM that follows the scan held the device (round 6: the read-back was serialised behind it).
__global__ void export_plan_kernel(const int32_t* __restrict__ max_count, const int32_t* __restrict__ fire_count,
                                   const int32_t* __restrict__ token_num, int B, volatile int32_t* dst) {
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    dst[1 + i] = fire_count[i];
    dst[1 + B + i] = token_num[i];
  }
  if (threadIdx.x == 0) dst[0] = max_count[0];
  __threadfence_system();
}
void launch_export_plan(hipStream_t s, const int32_t* max_count, const int32_t* fire_count, const int32_t* token_num, int B, int32_t* dst_dev) {
  hipLaunchKernelGGL(export_plan_kernel, dim3(1), dim3(256), 0, s, max_count, fire_count, token_num, B, dst_dev);
}

}  // namespace pf
