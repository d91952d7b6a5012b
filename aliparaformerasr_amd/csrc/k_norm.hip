// k_norm.hip — LayerNorm family (HBM-bound row kernels, one wavefront per row).
//
// Restates the LayerNormalization nodes of the FunASR ONNX graphs run by
// InferenceSession.Run (AliParaformerAsr/OfflineProjOfParaformer.cs:68): eps = 1e-12,
// statistics in fp32, two-pass (mean, then mean of squared deviations) so that the
// sentinel-padded rows (PadHelper.cs:63, |x| ~ 1.7e7 after the sqrt(512) scale) do not
// suffer E[x^2]-E[x]^2 cancellation.
//
// posenc_ln fuses the encoder prologue (x*sqrt(d_model) + sinusoidal PE, positions 1..T)
// into the first layer's norm1; the multiply and the add are kept as two separately
// rounded fp32 operations (no FMA contraction) to match a Mul node followed by an Add node.
#include "kernels.h"
#include "exact.h"

namespace pf {

#define LN_EPS 1e-12f

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// NV = float4 slots per lane; row width D (multiple of 4, D/4 <= 64*NV)
template <int NV, bool POSENC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t rows, int D,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        half_t* __restrict__ out16, int ld16,
                                                        float* __restrict__ out32, int ld32,
                                                        const float* __restrict__ pe, int T, float xscale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nq = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * (int64_t)D);
  const float4* per = nullptr;
  if (POSENC) per = reinterpret_cast<const float4*>(pe + (row % T) * (int64_t)D);
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      float4 t = xr[qd];
      if (POSENC) {
        const float4 p = per[qd];
        t.x = add_rn(mul_rn(t.x, xscale), p.x);
        t.y = add_rn(mul_rn(t.y, xscale), p.y);
        t.z = add_rn(mul_rn(t.z, xscale), p.z);
        t.w = add_rn(mul_rn(t.w, xscale), p.w);
      }
      v[i] = t;
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // Shifted statistics: d = x - x0 (x0 = first element of the row).  For the sentinel rows
  // (|x| ~ 1.7e7, spread of a few ulps) x - x0 is exact, so the mean and the deviations keep
  // full precision where sum(x)/D would not.
  const float x0 = __shfl(v[0].x, 0, 64);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      v[i].x = sub_rn(v[i].x, x0); v[i].y = sub_rn(v[i].y, x0);
      v[i].z = sub_rn(v[i].z, x0); v[i].w = sub_rn(v[i].w, x0);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;       // mean of (x - x0)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = wave_sum(ss) / (float)D;
  const float rstd = 1.0f / sqrtf(var + LN_EPS);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int qd = lane + 64 * i;
    if (qd < nq) {
      const float4 g = g4[qd], b = b4[qd];
      float4 y;
      y.x = (v[i].x - mean) * rstd * g.x + b.x;
      y.y = (v[i].y - mean) * rstd * g.y + b.y;
      y.z = (v[i].z - mean) * rstd * g.z + b.z;
      y.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (out32) reinterpret_cast<float4*>(out32 + row * (int64_t)ld32)[qd] = y;
      if (out16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 h = {(half_t)y.x, (half_t)y.y, (half_t)y.z, (half_t)y.w};
        reinterpret_cast<h4*>(out16 + row * (int64_t)ld16)[qd] = h;
      }
    }
  }
  // zero the K-padding columns D..ld16-1 of the f16 operand (layer 0: 560 -> 576)
  if (out16 && ld16 > D) {
    for (int c = D + lane; c < ld16; c += 64) out16[row * (int64_t)ld16 + c] = (half_t)0.f;
  }
}

template <bool POSENC>
static void ln_dispatch(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma, const float* beta,
                        half_t* out16, int ld16, float* out32, int ld32, const float* pe, int T, float xscale) {
  PF_CHECK(D % 4 == 0 && D <= 64 * 4 * 8, PF_ERR_INVALID_ARG, "layernorm: unsupported width");
  if (rows == 0) return;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nv = (D / 4 + 63) / 64;
#define LN_CASE(N)                                                                                         \
  case N:                                                                                                  \
    hipLaunchKernelGGL((layernorm_kernel<N, POSENC>), grid, block, 0, s, x, rows, D, gamma, beta, out16,   \
                       ld16, out32, ld32, pe, T, xscale);                                                  \
    break;
  switch (nv) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    default: throw Error(PF_ERR_INVALID_ARG, "layernorm: width too large");
  }
#undef LN_CASE
  PF_HIP(hipGetLastError());
}

void launch_layernorm(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma, const float* beta,
                      half_t* out16, int ld16, float* out32, int ld32) {
  ln_dispatch<false>(s, x, rows, D, gamma, beta, out16, ld16, out32, ld32, nullptr, 1, 1.f);
}

// pe: device table [>=T, F] built by the engine (SinusoidalPositionEncoder, positions 1..T)
void launch_posenc_ln_tab(hipStream_t s, const float* speech, int B, int T, int F, float xscale, const float* pe,
                          const float* gamma, const float* beta, half_t* out, int ldo) {
  ln_dispatch<true>(s, speech, (int64_t)B * T, F, gamma, beta, out, ldo, nullptr, 0, pe, T, xscale);
}

// fp32 -> f16 row copy (used for operands that arrive as fp32: stand-alone ops, CIF embeds)
__global__ void f32_to_f16_kernel(const float* __restrict__ x, int64_t rows, int cols, int ldx,
                                  half_t* __restrict__ y, int ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = rows * (int64_t)ldy;
  if (i >= total) return;
  const int64_t r = i / ldy;
  const int c = (int)(i - r * ldy);
  y[i] = c < cols ? (half_t)x[r * (int64_t)ldx + c] : (half_t)0.f;
}

void launch_f32_to_f16(hipStream_t s, const float* x, int64_t rows, int cols, int ldx, half_t* y, int ldy) {
  const int64_t total = rows * (int64_t)ldy;
  if (total == 0) return;
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, rows, cols,
                     ldx, y, ldy);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
