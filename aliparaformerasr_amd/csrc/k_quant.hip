// k_quant.hip — the quantisation steps of the int8 path (pf_engine_config.math_mode = 2).
//
// The reference's CLI default is `-accuracy int8` (AliParaformerAsr.Examples/Program.cs:98-101; model.int8.onnx,
// Examples/OfflineAliParaformerAsrRecognizer.cs:17-22): a FunASR export passed through onnxruntime's
// quantize_dynamic, in which every MatMul with a constant weight becomes
//     DynamicQuantizeLinear(x) -> MatMulInteger(x_q, w_q, x_zp, w_zp) -> Cast(float) -> Mul(x_scale * w_scale) [-> Add(bias)]
// (fused by onnxruntime into DynamicQuantizeMatMul / MatMulIntegerToFloat; same arithmetic).  Neither onnxruntime nor a
// model file is under /root/reference: the arithmetic below restates the ONNX operator definitions
// (DynamicQuantizeLinear-11: uint8, range widened to include 0, round-half-to-even, saturate) as onnxruntime's MLAS
// kernels implement them; oracle/int8.py is the numpy statement the op tests compare with, bit for bit.
//
//   activation (per TENSOR, dynamic):  min' = min(0, min x), max' = max(0, max x)
//       scale = max' == min' ? 1 : (max' - min') / 255          zp = rne(clamp(0 - min' / scale, 0, 255))
//       q     = clamp(rne(x / scale) + zp, 0, 255)
//   weight (per OUTPUT CHANNEL, static, quantize_dynamic(per_channel=True, weight_type=QUInt8)): the same formulas per row
//       of W [N, K].
// The matrix cores multiply SIGNED bytes, so both sides are stored minus 128 (a' = q - 128) together with the row /
// column sums that restore the exact integer sum (k_gemm.hip gemm_i8_pp3).
#include "kernels.h"
#include "exact.h"

namespace pf {

// monotone map float -> unsigned (so that atomicMin / atomicMax on the key order floats)
__device__ __forceinline__ unsigned fkey(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Pass 1 of DynamicQuantizeLinear: min / max of the tensor.  QMM_G workgroups of 16 waves (one per CU) each write ONE
// {min, max} pair (both start at 0: the range always includes 0) to part[2 b], part[2 b + 1]; pass 2 folds the QMM_G pairs
// (2 KB, L2-resident) at its start.  No atomics — a same-address atomic pair per workgroup serialised at ~11 ns each
// and was most of this pass (1024 workgroups: 22 us for a 32 MB tensor) — and nothing to re-arm between tensors.
constexpr int QMM_G = 256, QMM_T = 1024;

__device__ __forceinline__ void block_minmax_store(float lo, float hi, float* __restrict__ part) {
  lo = wave_min(lo);
  hi = wave_max(hi);
  __shared__ float s_lo[QMM_T / 64], s_hi[QMM_T / 64];
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int n = blockDim.x >> 6;
    lo = threadIdx.x < n ? s_lo[threadIdx.x] : 0.f;
    hi = threadIdx.x < n ? s_hi[threadIdx.x] : 0.f;
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = lo; part[2 * blockIdx.x + 1] = hi; }
  }
}
// the tensor's min / max from the QMM_G pairs (every lane gets both)
__device__ __forceinline__ void fold_minmax(const float* __restrict__ part, float& lo, float& hi) {
  const int lane = threadIdx.x & 63;
  lo = 0.f; hi = 0.f;
#pragma unroll
  for (int i = 0; i < QMM_G / 64; ++i) {
    const float2 v = reinterpret_cast<const float2*>(part)[lane + 64 * i];
    lo = fminf(lo, v.x);
    hi = fmaxf(hi, v.y);
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
}

template <bool F16>
__global__ __launch_bounds__(QMM_T) void minmax_kernel(const void* __restrict__ xv, int64_t rows, int cols, int ldx, float* __restrict__ part) {
  const int64_t cq = cols >> 2;
  const int64_t total = rows * cq;
  const int64_t stride = (int64_t)gridDim.x * QMM_T;
  float lo = 0.f, hi = 0.f;
  auto take = [&](int64_t i) __attribute__((always_inline)) -> float4 {
    const int64_t r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    if constexpr (F16) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      const h4 h = *reinterpret_cast<const h4*>(reinterpret_cast<const half_t*>(xv) + r * ldx + c);
      return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
      return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + r * ldx + c);
    }
  };
  auto fold = [&](float4 v) __attribute__((always_inline)) {
    lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
  };
  int64_t i = (int64_t)blockIdx.x * QMM_T + threadIdx.x;
  for (; i + 3 * stride < total; i += 4 * stride) {             // four independent loads in flight per lane
    const float4 a = take(i), b = take(i + stride), c = take(i + 2 * stride), d = take(i + 3 * stride);
    fold(a); fold(b); fold(c); fold(d);
  }
  for (; i < total; i += stride) fold(take(i));
  block_minmax_store(lo, hi, part);
}

__device__ __forceinline__ void qparams(float mn, float mx, float& scale, float& zp) {
#pragma clang fp contract(off)
  mn = fminf(mn, 0.f);
  mx = fmaxf(mx, 0.f);
  scale = mx == mn ? 1.0f : (mx - mn) / 255.0f;
  float z = 0.0f - mn / scale;
  z = fminf(fmaxf(z, 0.f), 255.f);
  zp = rintf(z);                                       // round half to even
}
__device__ __forceinline__ int qvalue(float x, float scale, float zp) {
#pragma clang fp contract(off)
  float v = rintf(x / scale) + zp;
  v = fminf(fmaxf(v, 0.f), 255.f);
  return (int)v - 128;                                  // the signed byte the matrix cores see
}

// one wave per row: q, a' = q - 128 packed four to a word, pad columns 0, row sum of a'
template <bool F16>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const void* __restrict__ xv, int64_t rows, int cols, int ldx, int8_t* __restrict__ out,
                                                            int ld, int32_t* __restrict__ rowsum, float* __restrict__ params,
                                                            const float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float mn, mx, scale, zp;
  fold_minmax(part, mn, mx);
  qparams(mn, mx, scale, zp);
  if (blockIdx.x == 0 && threadIdx.x == 0) { params[0] = scale; params[1] = zp; }
  if (r >= rows) return;
  int sum = 0;
  for (int c = lane * 4; c < ld; c += 256) {
    int q[4] = {0, 0, 0, 0};
    if (c < cols) {
      float v[4];
      if constexpr (F16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 h = *reinterpret_cast<const h4*>(reinterpret_cast<const half_t*>(xv) + r * ldx + c);
        v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
      } else {
        const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + r * ldx + c);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { q[e] = qvalue(v[e], scale, zp); sum += q[e]; }
    }
    const unsigned w = (unsigned)(q[0] & 255) | ((unsigned)(q[1] & 255) << 8) | ((unsigned)(q[2] & 255) << 16) | ((unsigned)(q[3] & 255) << 24);
    *reinterpret_cast<unsigned*>(out + r * (int64_t)ld + c) = w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) rowsum[r] = sum;
}

// The two passes of DynamicQuantizeLinear as separate launches; `part` = QMM_G {min, max} pairs (2 KB of scratch).
void launch_minmax(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, float* part) {
  PF_CHECK((x32 != nullptr) != (x16 != nullptr), PF_ERR_INVALID_ARG, "minmax: exactly one of x32 / x16");
  PF_CHECK(cols % 4 == 0 && ldx % 4 == 0, PF_ERR_INVALID_ARG, "minmax: cols / ldx must be multiples of 4");
  if (rows == 0) return;
  if (x16) hipLaunchKernelGGL(minmax_kernel<true>, dim3(QMM_G), dim3(QMM_T), 0, s, (const void*)x16, rows, cols, ldx, part);
  else hipLaunchKernelGGL(minmax_kernel<false>, dim3(QMM_G), dim3(QMM_T), 0, s, (const void*)x32, rows, cols, ldx, part);
  PF_HIP(hipGetLastError());
}
void launch_quantize(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, int8_t* out, int ld,
                     int32_t* rowsum, float* params, const float* part) {
  PF_CHECK((x32 != nullptr) != (x16 != nullptr), PF_ERR_INVALID_ARG, "quantize: exactly one of x32 / x16");
  PF_CHECK(cols % 4 == 0 && ldx % 4 == 0 && ld % 4 == 0 && ld >= cols, PF_ERR_INVALID_ARG, "quantize: cols / ldx / ld must be multiples of 4");
  if (rows == 0) return;
  const unsigned g2 = (unsigned)((rows + 3) / 4);
  if (x16) hipLaunchKernelGGL(quantize_rows_kernel<true>, dim3(g2), dim3(256), 0, s, (const void*)x16, rows, cols, ldx, out, ld, rowsum, params, part);
  else hipLaunchKernelGGL(quantize_rows_kernel<false>, dim3(g2), dim3(256), 0, s, (const void*)x32, rows, cols, ldx, out, ld, rowsum, params, part);
  PF_HIP(hipGetLastError());
}

// scratch: at least quant_scratch_bytes()
void launch_quantize_rows(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, int8_t* out, int ld,
                          int32_t* rowsum, float* params, unsigned* scratch) {
  if (rows == 0) return;
  launch_minmax(s, x32, x16, rows, cols, ldx, reinterpret_cast<float*>(scratch));
  launch_quantize(s, x32, x16, rows, cols, ldx, out, ld, rowsum, params, reinterpret_cast<const float*>(scratch));
}
size_t quant_scratch_bytes() { return (size_t)QMM_G * 2 * sizeof(float); }

// ---- LayerNorm fused with both passes: the normalised tensor never exists in memory.  Pass 1 normalises every row in
// registers and keeps only min / max; pass 2 normalises again (the same instructions: contraction off, so both passes see
// bit-identical values) and quantises.  x is read twice (2 x 32 MB at 16 000 x 512) instead of LayerNorm writing fp32
// (32 MB) that the min / max pass and the quantise pass each read back (64 MB).  Row arithmetic = k_norm.hip's shifted
// two-pass form (eps 1e-12; the sentinel rows of PadHelper.cs:63 keep their precision).
template <int NV>
__device__ __forceinline__ void ln_load(const float* __restrict__ xr, int D, int lane, float4 (&y)[NV]) {
  const int nq = D >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(xr);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int qd = lane + 64 * i;
    y[i] = qd < nq ? x4[qd] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NV>
__device__ __forceinline__ void ln_norm(int D, int lane, const float* __restrict__ gamma, const float* __restrict__ beta, float4 (&y)[NV]) {
#pragma clang fp contract(off)
  const int nq = D >> 2;
  const float x0 = __shfl(y[0].x, 0, 64);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nq) {
      y[i].x -= x0; y[i].y -= x0; y[i].z -= x0; y[i].w -= x0;
      s += (y[i].x + y[i].y) + (y[i].z + y[i].w);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nq) {
      const float a = y[i].x - mean, b = y[i].y - mean, c = y[i].z - mean, d = y[i].w - mean;
      ss += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rstd = 1.0f / sqrtf(ss / (float)D + 1e-12f);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      const float4 g = g4[qd], b = b4[qd];
      y[i].x = (y[i].x - mean) * rstd * g.x + b.x;
      y[i].y = (y[i].y - mean) * rstd * g.y + b.y;
      y[i].z = (y[i].z - mean) * rstd * g.z + b.z;
      y[i].w = (y[i].w - mean) * rstd * g.w + b.w;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(QMM_T) void ln_minmax_kernel(const float* __restrict__ x, int64_t rows, int D, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, nq = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * (QMM_T / 64);
  float lo = 0.f, hi = 0.f;
  int64_t r = (int64_t)blockIdx.x * (QMM_T / 64) + (threadIdx.x >> 6);
  float4 y[NV], nx[NV];
  if (r < rows) ln_load<NV>(x + r * (int64_t)D, D, lane, y);
  for (; r < rows; r += stride) {
    const bool more = r + stride < rows;
    if (more) ln_load<NV>(x + (r + stride) * (int64_t)D, D, lane, nx);     // the next row is in flight under this row's reductions
    ln_norm<NV>(D, lane, gamma, beta, y);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nq) {
        lo = fminf(fminf(lo, y[i].x), fminf(fminf(y[i].y, y[i].z), y[i].w));
        hi = fmaxf(fmaxf(hi, y[i].x), fmaxf(fmaxf(y[i].y, y[i].z), y[i].w));
      }
#pragma unroll
    for (int i = 0; i < NV; ++i) y[i] = nx[i];
  }
  block_minmax_store(lo, hi, part);
}

template <int NV>
__global__ __launch_bounds__(256) void ln_quantize_kernel(const float* __restrict__ x, int64_t rows, int D, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int8_t* __restrict__ out, int ld,
                                                          int32_t* __restrict__ rowsum, float* __restrict__ params,
                                                          const float* __restrict__ part) {
  const int lane = threadIdx.x & 63, nq = D >> 2;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float4 y[NV];
  if (r < rows) ln_load<NV>(x + r * (int64_t)D, D, lane, y);             // requested before the fold of the partial pairs
  float mn, mx, scale, zp;
  fold_minmax(part, mn, mx);
  qparams(mn, mx, scale, zp);
  if (blockIdx.x == 0 && threadIdx.x == 0) { params[0] = scale; params[1] = zp; }
  if (r >= rows) return;
  ln_norm<NV>(D, lane, gamma, beta, y);
  int sum = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < ld) {
      int q[4] = {0, 0, 0, 0};
      if (lane + 64 * i < nq) {
        q[0] = qvalue(y[i].x, scale, zp); q[1] = qvalue(y[i].y, scale, zp); q[2] = qvalue(y[i].z, scale, zp); q[3] = qvalue(y[i].w, scale, zp);
        sum += (q[0] + q[1]) + (q[2] + q[3]);
      }
      const unsigned w = (unsigned)(q[0] & 255) | ((unsigned)(q[1] & 255) << 8) | ((unsigned)(q[2] & 255) << 16) | ((unsigned)(q[3] & 255) << 24);
      *reinterpret_cast<unsigned*>(out + r * (int64_t)ld + c) = w;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) rowsum[r] = sum;
}

void launch_ln_minmax(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma, const float* beta, float* part) {
  PF_CHECK(D % 4 == 0 && D <= 2048, PF_ERR_INVALID_ARG, "ln_minmax: D must be a multiple of 4, at most 2048");
  if (rows == 0) return;
  if (D <= 512) hipLaunchKernelGGL(ln_minmax_kernel<2>, dim3(QMM_G), dim3(QMM_T), 0, s, x, rows, D, gamma, beta, part);
  else if (D <= 768) hipLaunchKernelGGL(ln_minmax_kernel<3>, dim3(QMM_G), dim3(QMM_T), 0, s, x, rows, D, gamma, beta, part);
  else hipLaunchKernelGGL(ln_minmax_kernel<8>, dim3(QMM_G), dim3(QMM_T), 0, s, x, rows, D, gamma, beta, part);
  PF_HIP(hipGetLastError());
}
void launch_ln_quantize(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma, const float* beta, int8_t* out, int ld,
                        int32_t* rowsum, float* params, const float* part) {
  PF_CHECK(D % 4 == 0 && ld % 4 == 0 && ld >= D && ld <= 2048, PF_ERR_INVALID_ARG, "ln_quantize: D / ld out of range");
  if (rows == 0) return;
  const unsigned g = (unsigned)((rows + 3) / 4);
  if (ld <= 512) hipLaunchKernelGGL(ln_quantize_kernel<2>, dim3(g), dim3(256), 0, s, x, rows, D, gamma, beta, out, ld, rowsum, params, part);
  else if (ld <= 768) hipLaunchKernelGGL(ln_quantize_kernel<3>, dim3(g), dim3(256), 0, s, x, rows, D, gamma, beta, out, ld, rowsum, params, part);
  else hipLaunchKernelGGL(ln_quantize_kernel<8>, dim3(g), dim3(256), 0, s, x, rows, D, gamma, beta, out, ld, rowsum, params, part);
  PF_HIP(hipGetLastError());
}

// one wave per output channel n: min / max of the row, its scale / zero point, the quantised row and its sum
__global__ __launch_bounds__(256) void quantize_weight_kernel(const float* __restrict__ W, int N, int K, int8_t* __restrict__ out, int ld,
                                                              int32_t* __restrict__ colsum, int32_t* __restrict__ wzp, float* __restrict__ wscale) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* w = W + (int64_t)n * K;
  float lo = 0.f, hi = 0.f;
  for (int k = lane; k < K; k += 64) { lo = fminf(lo, w[k]); hi = fmaxf(hi, w[k]); }
  lo = wave_min(lo);
  hi = wave_max(hi);
  float scale, zp;
  qparams(lo, hi, scale, zp);
  int sum = 0;
  for (int k = lane; k < ld; k += 64) {
    int q = 0;
    if (k < K) { q = qvalue(w[k], scale, zp); sum += q; }
    out[(int64_t)n * ld + k] = (int8_t)q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) { colsum[n] = sum; wzp[n] = (int)zp - 128; wscale[n] = scale; }
}

void launch_quantize_weight(hipStream_t s, const float* W, int N, int K, int8_t* out, int ld, int32_t* colsum, int32_t* wzp, float* wscale) {
  PF_CHECK(ld >= K, PF_ERR_INVALID_ARG, "quantize_weight: ld < K");
  if (N == 0) return;
  hipLaunchKernelGGL(quantize_weight_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, N, K, out, ld, colsum, wzp, wscale);
  PF_HIP(hipGetLastError());
}

// the stored bytes of an int8 export (container tensors `<linear>.weight_q` u8 [N, K], `.weight_zp` u8 [N], `.weight_scale`
// f32 [N]; aliparaformerasr_amd/convert.py) into the matrix-core operand: w' = q - 128, its column sums, zp - 128, scale.
__global__ __launch_bounds__(256) void import_weight_kernel(const uint8_t* __restrict__ Q, const uint8_t* __restrict__ zp,
                                                            const float* __restrict__ scale, int N, int K, int8_t* __restrict__ out, int ld,
                                                            int32_t* __restrict__ colsum, int32_t* __restrict__ wzp, float* __restrict__ wscale) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const uint8_t* q = Q + (int64_t)n * K;
  int sum = 0;
  for (int k = lane; k < ld; k += 64) {
    int v = 0;
    if (k < K) { v = (int)q[k] - 128; sum += v; }
    out[(int64_t)n * ld + k] = (int8_t)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) { colsum[n] = sum; wzp[n] = (int)zp[n] - 128; wscale[n] = scale[n]; }
}

void launch_import_weight(hipStream_t s, const uint8_t* Q, const uint8_t* zp, const float* scale, int N, int K, int8_t* out, int ld,
                          int32_t* colsum, int32_t* wzp, float* wscale) {
  PF_CHECK(ld >= K, PF_ERR_INVALID_ARG, "import_weight: ld < K");
  if (N == 0) return;
  hipLaunchKernelGGL(import_weight_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, Q, zp, scale, N, K, out, ld, colsum, wzp, wscale);
  PF_HIP(hipGetLastError());
}

// dz[n] = (colsum[n] - K wzp'[n]) * 256 + (wzp'[n] & 255): one word per column for the f16-result int8 GEMM's column line
__global__ void pack_dz_kernel(const int32_t* __restrict__ colsum, const int32_t* __restrict__ wzp, int N, int K, int32_t* __restrict__ dz) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) dz[n] = (colsum[n] - K * wzp[n]) * 256 + (wzp[n] & 255);
}
void launch_pack_dz(hipStream_t s, const int32_t* colsum, const int32_t* wzp, int N, int K, int32_t* dz) {
  PF_CHECK(K <= 16384, PF_ERR_UNSUPPORTED, "pack_dz: K too large for the packed column word");
  if (N == 0) return;
  hipLaunchKernelGGL(pack_dz_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, colsum, wzp, N, K, dz);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
