// k_bicif.hip — BiCIF timestamp head (CifPredictorV3.get_upsample_timestmap of the FunASR export;
// its output `us_cif_peak` is graph output [3] consumed by AliParaformerAsr/OfflineRecognizer.cs:172-183
// and turned into timestamps by time_stamp_lfr6_onnx, OfflineRecognizer.cs:200-302).
//
//   up   = ConvTranspose1d(D, D, k=3, stride=3)(H)           -> GEMM  [M,512] x [512,1536]   (k_gemm.hip)
//   xg   = up W_ih^T + b_ih + b_hh  (both directions)         -> GEMM  [3M,512] x [512,4096]  (k_gemm.hip)
//   h_t  = LSTM cell, forward and reverse direction            -> lstm_step_kernel, one launch per time step
//   a2   = relu(sigmoid(h W_o^T + b_o) * smooth2 - noise2)     -> us_alpha_kernel
//   a2  *= token_num / sum(a2);  peak = running integrate       -> us_peak_kernel
//
// The recurrence is a chain of 3T dependent steps; round 1 runs it as one small launch per step
// (latency bound, ~5 us/step).  Each launch covers both directions; a block owns 8 hidden units
// (x 4 gates = one 32-row MFMA tile of W_hh) for a tile of 32 utterances, its 4 waves split K = 512.
#include <hip/hip_runtime.h>

#include "exact.h"
#include "kernels.h"

namespace pf {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Row rho of the MFMA A tile is W_hh row (gate = rho & 3, unit = u0 + (rho >> 2)); the 32x32 result
// then leaves every lane with the 4 gates of 4 units for one utterance (D rows 8q + 4*(lane>>5) + r).
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmArgs a) {
  const int D = a.D;
  const int ub = blockIdx.x, dir = blockIdx.y, bt = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = dir == 0 ? a.step : a.T3 - 1 - a.step;
  const int pp = a.step & 1;
  const half_t* hprev = a.hstate + (size_t)(dir * 2 + pp) * a.B * D;
  half_t* hnext = a.hstate + (size_t)(dir * 2 + (pp ^ 1)) * a.B * D;
  const int r = lane & 31, kg = lane >> 5;
  const half_t* wrow = a.whh + ((size_t)dir * 4 * D + (size_t)(r & 3) * D + ub * 8 + (r >> 2)) * D;
  const int bb = min(bt * 32 + r, a.B - 1);
  const half_t* hrow = hprev + (size_t)bb * D;

  // the cell's own inputs (this thread's unit / utterance) are fetched first: their HBM latency then
  // overlaps the W_hh / h loads and the MFMAs instead of following the LDS reduction
  const int bq = min(bt * 32 + r, a.B - 1);
  const int uq = ub * 8 + 2 * wave + kg;
  const float* xgp = a.xg + ((size_t)bq * a.T3 + t) * (size_t)(a.ndir * 4 * D) + (size_t)dir * 4 * D + uq;
  float* cp = a.cstate + ((size_t)dir * a.B + bq) * D + uq;
  const float xi = xgp[0], xf = xgp[D], xc = xgp[2 * D], xo = xgp[3 * D], cprev = *cp;

  const int kspan = D / 4;               // K slice of this wave
  f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = wave * kspan; k0 < (wave + 1) * kspan; k0 += 128) {
    h8v av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      av[s] = *reinterpret_cast<const h8v*>(wrow + k0 + s * 16 + kg * 8);
      bv[s] = *reinterpret_cast<const h8v*>(hrow + k0 + s * 16 + kg * 8);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bv[s], acc, 0, 0, 0);
  }

  __shared__ float red[4][16][64];
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  float g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    g[q] = (red[0][4 * wave + q][lane] + red[1][4 * wave + q][lane]) + (red[2][4 * wave + q][lane] + red[3][4 * wave + q][lane]);

  const int b = bt * 32 + r;
  if (b >= a.B) return;
  const int u = uq;
  const float gi = g[0] + xi, gf = g[1] + xf, gg = g[2] + xc, go = g[3] + xo;
  const float c = sigmoidf_(gf) * cprev + sigmoidf_(gi) * tanhf(gg);
  const float h = sigmoidf_(go) * tanhf(c);
  *cp = c;
  hnext[(size_t)b * D + u] = (half_t)h;
  a.hout[((size_t)b * a.T3 + t) * (size_t)(a.ndir * D) + (size_t)dir * D + u] = h;
}

void launch_lstm_step(hipStream_t s, const LstmArgs& a) {
  PF_CHECK(a.D % 512 == 0, PF_ERR_UNSUPPORTED, "lstm: hidden size must be a multiple of 512");
  PF_CHECK(a.ndir == 1 || a.ndir == 2, PF_ERR_INVALID_ARG, "lstm: ndir must be 1 or 2");
  hipLaunchKernelGGL(lstm_step_kernel, dim3(a.D / 8, a.ndir, cdiv(a.B, 32)), dim3(256), 0, s, a);
  PF_HIP(hipGetLastError());
}

// a2raw[row] = relu(sigmoid(dot(hout[row, 0:W], w) + b0) * smooth - noise); one wave per row
__global__ __launch_bounds__(256) void us_alpha_kernel(const float* __restrict__ hout, int64_t rows, int W,
                                                       const float* __restrict__ w, const float* __restrict__ b0,
                                                       float smooth, float noise, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = hout + row * W;
  float s = 0.f;
  for (int c = lane * 4; c < W; c += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + c);
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const float sg = sigmoidf_(s + b0[0]);
    out[row] = fmaxf(sub_rn(mul_rn(sg, smooth), noise), 0.f);
  }
}

// per utterance: alphas *= token_num / sum(alphas) (sum carried in double), then cif_wo_hidden with
// threshold thr: integrate += alpha; peak[t] = integrate; integrate -= thr once it reaches thr.
__global__ __launch_bounds__(64) void us_peak_kernel(float* __restrict__ alphas, const int32_t* __restrict__ token_num,
                                                     int T3, float thr, float* __restrict__ peak) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* a = alphas + (int64_t)b * T3;
  float* p = peak + (int64_t)b * T3;
  double s = 0.0;
  for (int t = lane; t < T3; t += 64) s += (double)a[t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float sum = (float)s;
  const float ratio = (float)token_num[b] / sum;
  for (int t = lane; t < T3; t += 64) a[t] = mul_rn(a[t], ratio);
  __threadfence_block();
  __syncthreads();
  if (lane != 0) return;
  float integrate = 0.f;
  for (int t = 0; t < T3; ++t) {
    integrate = add_rn(integrate, a[t]);
    p[t] = integrate;
    if (integrate >= thr) integrate = sub_rn(integrate, thr);
  }
}

void launch_us_alpha(hipStream_t s, const float* hout, int64_t rows, int W, const float* w, const float* b0,
                     float smooth, float noise, float* out) {
  if (rows == 0) return;
  PF_CHECK(W % 4 == 0, PF_ERR_UNSUPPORTED, "us_alpha: width must be a multiple of 4");
  hipLaunchKernelGGL(us_alpha_kernel, dim3((unsigned)cdiv(rows, (int64_t)4)), dim3(256), 0, s, hout, rows, W, w, b0,
                     smooth, noise, out);
  PF_HIP(hipGetLastError());
}

void launch_us_peak(hipStream_t s, float* alphas, const int32_t* token_num, int B, int T3, float thr, float* peak) {
  if (B == 0) return;
  hipLaunchKernelGGL(us_peak_kernel, dim3(B), dim3(64), 0, s, alphas, token_num, T3, thr, peak);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
