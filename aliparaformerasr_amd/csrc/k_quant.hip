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

// scratch[0] = key(min), scratch[1] = key(max), both initialised to key(0.0f): the range always includes 0
template <bool F16>
__global__ __launch_bounds__(256) void minmax_kernel(const void* __restrict__ xv, int64_t rows, int cols, int ldx, unsigned* __restrict__ scratch) {
  const int64_t cq = cols >> 2;
  const int64_t total = rows * cq;
  float lo = 0.f, hi = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    float4 v;
    if constexpr (F16) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      const h4 h = *reinterpret_cast<const h4*>(reinterpret_cast<const half_t*>(xv) + r * ldx + c);
      v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
      v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + r * ldx + c);
    }
    lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  // one atomic pair per WORKGROUP: same-address atomics serialise at ~11 ns each (8192 of them cost 90 us per tensor)
  __shared__ float s_lo[4], s_hi[4];
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(scratch, fkey(fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]))));
    atomicMax(scratch + 1, fkey(fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]))));
  }
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
                                                            const unsigned* __restrict__ scratch) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float scale, zp;
  qparams(fkey_inv(scratch[0]), fkey_inv(scratch[1]), scale, zp);
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

void launch_quantize_rows(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, int8_t* out, int ld,
                          int32_t* rowsum, float* params, unsigned* scratch) {
  PF_CHECK((x32 != nullptr) != (x16 != nullptr), PF_ERR_INVALID_ARG, "quantize_rows: exactly one of x32 / x16");
  PF_CHECK(cols % 4 == 0 && ldx % 4 == 0 && ld % 4 == 0 && ld >= cols, PF_ERR_INVALID_ARG, "quantize_rows: cols / ldx / ld must be multiples of 4");
  if (rows == 0) return;
  PF_HIP(hipMemsetD32Async((hipDeviceptr_t)scratch, (int)0x80000000u, 2, s));     // key(0.0f) twice
  const int64_t total = rows * (cols / 4);
  const unsigned g1 = (unsigned)std::min<int64_t>((total + 255) / 256, 512);
  const unsigned g2 = (unsigned)((rows + 3) / 4);
  if (x16) {
    hipLaunchKernelGGL(minmax_kernel<true>, dim3(g1), dim3(256), 0, s, (const void*)x16, rows, cols, ldx, scratch);
    hipLaunchKernelGGL(quantize_rows_kernel<true>, dim3(g2), dim3(256), 0, s, (const void*)x16, rows, cols, ldx, out, ld, rowsum, params, scratch);
  } else {
    hipLaunchKernelGGL(minmax_kernel<false>, dim3(g1), dim3(256), 0, s, (const void*)x32, rows, cols, ldx, scratch);
    hipLaunchKernelGGL(quantize_rows_kernel<false>, dim3(g2), dim3(256), 0, s, (const void*)x32, rows, cols, ldx, out, ld, rowsum, params, scratch);
  }
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

}  // namespace pf
