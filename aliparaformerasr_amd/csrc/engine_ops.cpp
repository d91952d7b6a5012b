// engine_ops.cpp — stand-alone operator entry points of the engine (pf_op_*: what the parity tests and the tools/ benches call
// to run ONE kernel form on caller data).  Split out of engine.cpp in round 6.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>

#include "hostutil.h"

namespace pf {

static const size_t kAlign = 256;

// ------------------------------------------------------------------ stand-alone ops -------
void Engine::op_lfr_cmvn_pad(const float* const* fbank, const int32_t* t80, int B, int sentinel, float* out,
                             int64_t cap, int32_t* tmax_out) {
  PF_HIP(hipSetDevice(device_));
  const int nm = fc_.n_mels, W = fc_.lfr_m * nm;
  std::vector<int64_t> foff(B + 1, 0);
  int tmax = 0;
  for (int b = 0; b < B; ++b) { foff[b + 1] = foff[b] + t80[b]; tmax = std::max(tmax, t80[b] / fc_.lfr_n); }
  if (tmax_out) *tmax_out = tmax;
  const int64_t need = (int64_t)B * tmax * W;
  PF_CHECK(cap >= need, PF_ERR_CAPACITY, "lfr_cmvn_pad: out capacity < " + std::to_string(need));
  if (need == 0) return;
  ensure(ws_fbank_, (size_t)std::max<int64_t>(foff[B], 1) * nm * 4);
  ensure(ws_meta_, (size_t)(B + 1) * 8 + (size_t)B * 4 + 64);
  ensure(ws_speech_, (size_t)need * 4);
  for (int b = 0; b < B; ++b)
    if (t80[b] > 0)
      PF_HIP(hipMemcpyAsync((float*)ws_fbank_.p + foff[b] * nm, fbank[b], (size_t)t80[b] * nm * 4,
                            hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(ws_meta_.p, foff.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream_));
  int32_t* t80d = (int32_t*)((char*)ws_meta_.p + (size_t)(B + 1) * 8);
  PF_HIP(hipMemcpyAsync(t80d, t80, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
  launch_lfr_cmvn_pad(stream_, (const float*)ws_fbank_.p, (const int64_t*)ws_meta_.p, t80d, B, tmax, fc_.lfr_m,
                      fc_.lfr_n, nm, cmvn_shift_, cmvn_scale_, cmvn_shift_ ? 1 : 0, sentinel, (float*)ws_speech_.p);
  PF_HIP(hipMemcpyAsync(out, ws_speech_.p, (size_t)need * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_argmax(const float* x, int64_t rows, int V, int64_t* ids) {
  PF_HIP(hipSetDevice(device_));
  if (rows == 0) return;
  ensure(ws_tmp_, (size_t)rows * V * 4 + (size_t)rows * 8 + 256);
  float* xd = (float*)ws_tmp_.p;
  int64_t* idd = (int64_t*)((char*)ws_tmp_.p + round_up((int64_t)rows * V * 4, 256));
  PF_HIP(hipMemcpyAsync(xd, x, (size_t)rows * V * 4, hipMemcpyHostToDevice, stream_));
  launch_argmax(stream_, xd, rows, V, V, 0, idd);
  PF_HIP(hipMemcpyAsync(ids, idd, (size_t)rows * 8, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_gemm(const float* A, const float* W, const float* bias, int M, int N, int K, int epi, float* C) {
  PF_HIP(hipSetDevice(device_));
  if (M == 0 || N == 0) return;
  const int Kp = (int)round_up(K, 64);
  const int64_t Mp = round_up(M, 256), Np = round_up(N, 256);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t oA = carve((size_t)M * K * 4), oW = carve((size_t)N * K * 4), ob = carve((size_t)N * 4);
  const size_t oA16 = carve((size_t)Mp * Kp * 2), oW16 = carve((size_t)Np * Kp * 2), oC = carve((size_t)Mp * N * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemcpyAsync(base + oA, A, (size_t)M * K * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + oW, W, (size_t)N * K * 4, hipMemcpyHostToDevice, stream_));
  if (bias) PF_HIP(hipMemcpyAsync(base + ob, bias, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemsetAsync(base + oA16, 0, (size_t)Mp * Kp * 2, stream_));
  PF_HIP(hipMemsetAsync(base + oW16, 0, (size_t)Np * Kp * 2, stream_));
  launch_f32_to_f16(stream_, (const float*)(base + oA), M, K, K, (half_t*)(base + oA16), Kp);
  launch_f32_to_f16(stream_, (const float*)(base + oW), N, K, K, (half_t*)(base + oW16), Kp);
  GemmArgs g{};
  g.A = (half_t*)(base + oA16); g.lda = Kp; g.W = (half_t*)(base + oW16); g.ldw = Kp;
  g.bias = bias ? (const float*)(base + ob) : nullptr;
  g.M = M; g.N = N; g.K = Kp; g.relu = epi == 1;
  g.out_padded = 1;
  const bool f16out = epi == 2;
  if (f16out) { g.out_f16 = (half_t*)(base + oC); g.ldc16 = N; }
  else { g.out_f32 = (float*)(base + oC); g.ldc32 = N; }
  prof_begin("gemm_op", 2.0 * M * (double)N * K);
  launch_gemm(stream_, g);
  prof_end("gemm_op");
  if (f16out) {
    std::vector<uint16_t> tmp((size_t)M * N);
    PF_HIP(hipMemcpyAsync(tmp.data(), base + oC, tmp.size() * 2, hipMemcpyDeviceToHost, stream_));
    PF_HIP(hipStreamSynchronize(stream_));
    for (size_t i = 0; i < tmp.size(); ++i) {
      half_t hv;
      std::memcpy(&hv, &tmp[i], 2);
      C[i] = (float)hv;
    }
  } else {
    PF_HIP(hipMemcpyAsync(C, base + oC, (size_t)M * N * 4, hipMemcpyDeviceToHost, stream_));
    PF_HIP(hipStreamSynchronize(stream_));
  }
}

void Engine::x3_forget(const float* W) {
  for (auto it = x3w_.begin(); it != x3w_.end();) {
    if (it->first.first != W) { ++it; continue; }
    for (size_t i = 0; i < owned_.size(); ++i)
      if (owned_[i] == (void*)it->second) { owned_.erase(owned_.begin() + i); break; }
    hipFree(it->second);
    it = x3w_.erase(it);
  }
}

// A Linear / the FFN block of the fp32 graph exactly as enc_layer_fp32() launches them (gemm32: math_mode 1 on the fp32
// matrix path, math_mode 3 as x3 products with the K-loop wrap and, in the block, the operand-pair epilogue).
void Engine::op_linear32(const float* x, const float* W, const float* bias, const float* resid, int M, int N, int K, bool relu, float* y) {
  PF_CHECK(fp32_mode_, PF_ERR_UNSUPPORTED, "linear32: the engine was not created with math_mode 1 or 3");
  PF_HIP(hipSetDevice(device_));
  const int ldc = (int)round_up(N, 4);
  const int64_t Mp = round_up(M, 256) + 128, Np = round_up(N, 256);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t ox = carve((size_t)Mp * K * 4), oW = carve((size_t)Np * K * 4), ob = carve((size_t)Np * 4), orr = carve((size_t)Mp * ldc * 4),
               oy = carve((size_t)Mp * ldc * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base, 0, off, stream_));
  PF_HIP(hipMemcpyAsync(base + ox, x, (size_t)M * K * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + oW, W, (size_t)N * K * 4, hipMemcpyHostToDevice, stream_));
  if (bias) PF_HIP(hipMemcpyAsync(base + ob, bias, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
  if (resid) PF_HIP(hipMemcpy2DAsync(base + orr, (size_t)ldc * 4, resid, (size_t)N * 4, (size_t)N * 4, M, hipMemcpyHostToDevice, stream_));
  gemm32((const float*)(base + ox), K, (const float*)(base + oW), K, bias ? (const float*)(base + ob) : nullptr, M, N, K, (float*)(base + oy), ldc,
         resid ? (const float*)(base + orr) : nullptr, ldc, relu, 0, 1.f);
  PF_HIP(hipMemcpy2DAsync(y, (size_t)N * 4, base + oy, (size_t)ldc * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  x3_forget((const float*)(base + oW));
  x3a_src_ = nullptr;
}

void Engine::op_ffn32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, int M, int D, int F, float* y) {
  PF_CHECK(fp32_mode_, PF_ERR_UNSUPPORTED, "ffn32: the engine was not created with math_mode 1 or 3");
  PF_CHECK(D % 4 == 0 && F % 4 == 0, PF_ERR_INVALID_ARG, "ffn32: D and F must be multiples of 4");
  PF_HIP(hipSetDevice(device_));
  const int64_t Mp = round_up(M, 256) + 128, Dp = round_up(D, 256), Fp = round_up(F, 256);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t ox = carve((size_t)Mp * D * 4), o1 = carve((size_t)Fp * D * 4), ob1 = carve((size_t)Fp * 4), o2 = carve((size_t)Dp * F * 4),
               ob2 = carve((size_t)Dp * 4), oh = carve((size_t)Mp * F * 4), oy = carve((size_t)Mp * D * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base, 0, off, stream_));
  PF_HIP(hipMemcpyAsync(base + ox, x, (size_t)M * D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + o1, W1, (size_t)F * D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ob1, b1, (size_t)F * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + o2, W2, (size_t)D * F * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ob2, b2, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
  const float* xd = (const float*)(base + ox);
  gemm32(xd, D, (const float*)(base + o1), D, (const float*)(base + ob1), M, F, D, (float*)(base + oh), F, nullptr, 0, true, 0, 1.f, kX3OutPair);
  gemm32((const float*)(base + oh), F, (const float*)(base + o2), F, (const float*)(base + ob2), M, D, F, (float*)(base + oy), D, xd, D, false, 0, 1.f,
         kX3InPair);
  PF_HIP(hipMemcpyAsync(y, base + oy, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  x3_forget((const float*)(base + o1));
  x3_forget((const float*)(base + o2));
  x3a_src_ = nullptr;
}

// GEMM exactly as the pipeline launches it: kernel kind (fp32 / f16 row-major / f16 blocked result), tile height,
// blocked A operand, residual / second addend, column scaling — the stand-alone counterpart of Engine::gemm().
void Engine::op_gemm_ex(const pf_gemm_desc& ds, const float* A, const float* W, float* C) {
  PF_HIP(hipSetDevice(device_));
  const int M = ds.M, N = ds.N, K = ds.K;
  PF_CHECK(M > 0 && N > 0 && K > 0, PF_ERR_INVALID_ARG, "gemm_ex: empty problem");
  PF_CHECK(ds.out_kind >= 0 && ds.out_kind <= 2, PF_ERR_INVALID_ARG, "gemm_ex: out_kind must be 0, 1 or 2");
  PF_CHECK(ds.tile_rows == 0 || ds.tile_rows == 32 || ds.tile_rows == 128 || ds.tile_rows == 256 || ds.tile_rows == 512 || ds.tile_rows == 1024 ||
               ds.tile_rows == 2048,
           PF_ERR_INVALID_ARG, "gemm_ex: tile_rows must be 0, 32 (= the short-input kernel), 128, 256, 512 (= the 256 x {192,256} tile kernel), "
           "1024 (= its persistent form for the blocked result) or 2048 (= the k-step-32 fp32-result kernel)");
  PF_CHECK(ds.out_kind == 0 || (!ds.resid && !ds.add2), PF_ERR_INVALID_ARG, "gemm_ex: residual / addend need the fp32 result kind");
  PF_CHECK(ds.out_kind != 2 || N % 64 == 0, PF_ERR_INVALID_ARG, "gemm_ex: blocked result needs N % 64 == 0");
  const int Kp = (int)round_up(K, 64);
  const int64_t Mp = round_up(M, 256) + 128, Np = round_up(N, 256);
  const int ld32 = (int)round_up(N, 4), ld16 = (int)round_up(N, 8);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t oA = carve((size_t)M * K * 4), oW = carve((size_t)N * K * 4), ob = carve((size_t)N * 4);
  const size_t oA16 = carve((size_t)Mp * Kp * 2), oW16 = carve((size_t)Np * Kp * 2);
  const size_t oR = carve((size_t)Mp * ld32 * 4), oD = carve((size_t)Mp * ld32 * 4), oC = carve((size_t)Mp * std::max(ld32, ld16) * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + oA16, 0, (size_t)Mp * Kp * 2, stream_));
  PF_HIP(hipMemsetAsync(base + oW16, 0, (size_t)Np * Kp * 2, stream_));
  std::vector<half_t> ablk;
  if (ds.a_blocked) {
    // host-side re-layout (independent of the kernel's own index arithmetic): element (m, k) at
    // ((m/32 * Kp/8 + k/8) * 32 + m%32) * 8 + k%8
    ablk.assign((size_t)Mp * Kp, (half_t)0.f);
    for (int m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k)
        ablk[(((size_t)(m >> 5) * (Kp >> 3) + (k >> 3)) * 32 + (m & 31)) * 8 + (k & 7)] = (half_t)A[(size_t)m * K + k];
    PF_HIP(hipMemcpyAsync(base + oA16, ablk.data(), ablk.size() * 2, hipMemcpyHostToDevice, stream_));
  } else {
    PF_HIP(hipMemcpyAsync(base + oA, A, (size_t)M * K * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + oA), M, K, K, (half_t*)(base + oA16), Kp);
  }
  PF_HIP(hipMemcpyAsync(base + oW, W, (size_t)N * K * 4, hipMemcpyHostToDevice, stream_));
  launch_f32_to_f16(stream_, (const float*)(base + oW), N, K, K, (half_t*)(base + oW16), Kp);
  if (ds.bias) PF_HIP(hipMemcpyAsync(base + ob, ds.bias, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
  if (ds.resid)
    PF_HIP(hipMemcpy2DAsync(base + oR, (size_t)ld32 * 4, ds.resid, (size_t)N * 4, (size_t)N * 4, M, hipMemcpyHostToDevice, stream_));
  if (ds.add2)
    PF_HIP(hipMemcpy2DAsync(base + oD, (size_t)ld32 * 4, ds.add2, (size_t)N * 4, (size_t)N * 4, M, hipMemcpyHostToDevice, stream_));
  GemmArgs g{};
  g.A = (half_t*)(base + oA16); g.lda = Kp; g.W = (half_t*)(base + oW16); g.ldw = Kp;
  g.bias = ds.bias ? (const float*)(base + ob) : nullptr;
  g.M = M; g.N = N; g.K = Kp; g.relu = ds.relu ? 1 : 0;
  g.scale_cols = ds.scale_cols; g.scale = ds.scale;
  g.out_padded = 1;
  g.a_blocked = ds.a_blocked ? 1 : 0;
  g.force_mi = ds.tile_rows == 128 ? 1 : (ds.tile_rows == 256 ? 2 : (ds.tile_rows == 32 ? 4 : (ds.tile_rows == 1024 ? 5 : (ds.tile_rows == 2048 ? 6 : 0))));
  g.small_ws = small_ws_;
  if (ds.out_kind == 0) {
    g.out_f32 = (float*)(base + oC); g.ldc32 = ld32;
    if (ds.resid) { g.resid = (const float*)(base + oR); g.ldr = ld32; }
    if (ds.add2) { g.add2 = (const float*)(base + oD); g.ld2 = ld32; }
  } else {
    g.out_f16 = (half_t*)(base + oC); g.ldc16 = ds.out_kind == 2 ? N : ld16;
    g.out_blocked = ds.out_kind == 2;
  }
  static const int reps = [] { const char* e = getenv("PF_OP_REPEAT"); return e ? std::max(1, atoi(e)) : 1; }();   // tools/: warm-cache timing
  for (int r = 0; r < reps; ++r) {
    prof_begin(r == 0 ? "gemm_op" : "gemm_op_warm", 2.0 * M * (double)N * K);
    launch_gemm(stream_, g);
    prof_end(r == 0 ? "gemm_op" : "gemm_op_warm");
  }
  if (ds.out_kind == 0) {
    PF_HIP(hipMemcpy2DAsync(C, (size_t)N * 4, base + oC, (size_t)ld32 * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost, stream_));
    PF_HIP(hipStreamSynchronize(stream_));
  } else {
    const size_t rows = ds.out_kind == 2 ? (size_t)round_up(M, 32) : (size_t)M;
    const size_t ld = ds.out_kind == 2 ? (size_t)N : (size_t)ld16;
    std::vector<half_t> tmp(rows * ld);
    PF_HIP(hipMemcpyAsync(tmp.data(), base + oC, tmp.size() * 2, hipMemcpyDeviceToHost, stream_));
    PF_HIP(hipStreamSynchronize(stream_));
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        const size_t idx = ds.out_kind == 2 ? (((size_t)(m >> 5) * (N >> 3) + (n >> 3)) * 32 + (m & 31)) * 8 + (n & 7)
                                            : (size_t)m * ld + n;
        C[(size_t)m * N + n] = (float)tmp[idx];
      }
  }
}

void Engine::op_gemm_rc(const pf_gemm_rc_desc& ds, const float* A, const float* W, float* x_out, float* n16_out,
                        float* n32_out) {
  PF_HIP(hipSetDevice(device_));
  const int M = ds.M, K = ds.K, N = 512;
  PF_CHECK(M > 0 && K > 0 && K % 64 == 0, PF_ERR_INVALID_ARG, "gemm_rc: K must be a positive multiple of 64");
  PF_CHECK(!ds.fsmn_v || (ds.fsmn_w && ds.fsmn_k > 0), PF_ERR_INVALID_ARG, "gemm_rc: FSMN needs weights");
  PF_CHECK((ds.ln_gamma != nullptr) == (ds.ln_beta != nullptr), PF_ERR_INVALID_ARG, "gemm_rc: gamma and beta go together");
  const int64_t Mp = round_up(M, 256) + 128;
  const int k = ds.fsmn_k;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o32 = carve((size_t)std::max<int64_t>((int64_t)M * std::max(K, N), (int64_t)N * K) * 4);
  const size_t oA16 = carve((size_t)Mp * K * 2), oW16 = carve((size_t)N * K * 2), ob = carve((size_t)N * 4);
  const size_t oR = carve((size_t)Mp * N * 4), oV = carve((size_t)(Mp + 128) * 3 * N * 2), owT = carve((size_t)std::max(k, 1) * N * 4);
  const size_t og = carve((size_t)N * 4), obe = carve((size_t)N * 4), oX = carve((size_t)Mp * N * 4), oN16 = carve((size_t)Mp * N * 2);
  const size_t oN32 = carve((size_t)Mp * N * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + oA16, 0, (size_t)Mp * K * 2, stream_));
  std::vector<half_t> ablk;
  if (ds.a_blocked) {
    ablk.assign((size_t)Mp * K, (half_t)0.f);
    for (int m = 0; m < M; ++m)
      for (int kk = 0; kk < K; ++kk)
        ablk[(((size_t)(m >> 5) * (K >> 3) + (kk >> 3)) * 32 + (m & 31)) * 8 + (kk & 7)] = (half_t)A[(size_t)m * K + kk];
    PF_HIP(hipMemcpyAsync(base + oA16, ablk.data(), ablk.size() * 2, hipMemcpyHostToDevice, stream_));
  } else {
    PF_HIP(hipMemcpyAsync(base + o32, A, (size_t)M * K * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + o32), M, K, K, (half_t*)(base + oA16), K);
    PF_HIP(hipStreamSynchronize(stream_));
  }
  PF_HIP(hipMemcpyAsync(base + o32, W, (size_t)N * K * 4, hipMemcpyHostToDevice, stream_));
  launch_f32_to_f16(stream_, (const float*)(base + o32), N, K, K, (half_t*)(base + oW16), K);
  PF_HIP(hipStreamSynchronize(stream_));
  GemmRcArgs g{};
  g.A = (half_t*)(base + oA16); g.lda = K; g.a_blocked = ds.a_blocked ? 1 : 0;
  g.W = (half_t*)(base + oW16); g.ldw = K; g.M = M; g.K = K;
  if (ds.bias) { PF_HIP(hipMemcpyAsync(base + ob, ds.bias, (size_t)N * 4, hipMemcpyHostToDevice, stream_)); g.bias = (const float*)(base + ob); }
  if (ds.resid) { PF_HIP(hipMemcpyAsync(base + oR, ds.resid, (size_t)M * N * 4, hipMemcpyHostToDevice, stream_)); g.resid = (const float*)(base + oR); g.ldr = N; }
  std::vector<float> wT;
  if (ds.fsmn_v) {
    // the V slice of a [M, 3*512] QKV buffer, as in the pipeline (row stride 1536 halves)
    PF_HIP(hipMemsetAsync(base + oV, 0, (size_t)(Mp + 128) * 3 * N * 2, stream_));
    PF_HIP(hipMemcpyAsync(base + o32, ds.fsmn_v, (size_t)M * N * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + o32), M, N, N, (half_t*)(base + oV) + 2 * N, 3 * N);
    wT.resize((size_t)k * N);
    for (int c = 0; c < N; ++c)
      for (int j = 0; j < k; ++j) wT[(size_t)j * N + c] = ds.fsmn_w[(size_t)c * k + j];
    PF_HIP(hipMemcpyAsync(base + owT, wT.data(), wT.size() * 4, hipMemcpyHostToDevice, stream_));
    g.fsmn_v = (half_t*)(base + oV) + 2 * N; g.ldv = 3 * N; g.fsmn_wT = (const float*)(base + owT); g.fsmn_k = k;
  }
  g.T = ds.T > 0 ? ds.T : M;
  if (ds.ln_gamma) {
    PF_HIP(hipMemcpyAsync(base + og, ds.ln_gamma, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
    PF_HIP(hipMemcpyAsync(base + obe, ds.ln_beta, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
    g.ln_g = (const float*)(base + og); g.ln_b = (const float*)(base + obe); g.eps = 1e-12f;
    if (n16_out) { g.out_n16 = (half_t*)(base + oN16); g.ldn16 = N; }
    if (n32_out) { g.out_n32 = (float*)(base + oN32); g.ldn32 = N; }
  }
  if (x_out) { g.out_x = (float*)(base + oX); g.ldx = N; }
  PF_CHECK(!ds.split_k, PF_ERR_UNSUPPORTED, "gemm_rc: the split-K forms (k_gemm_sk.hip) were removed in round 5 — the fused FFN block replaced them (numbers: profiles/round4_splitk_pairs.txt)");
  if (ds.short_input) {
    // the short-input forms of the same nodes, as enc_layer / the decoder run them for M <= 512 rows
    PF_CHECK(!ds.a_blocked, PF_ERR_INVALID_ARG, "gemm_rc: the short-input kernels take a row-major A");
    GemmSmallArgs q{};
    q.A = g.A; q.lda = K; q.W = g.W; q.ldw = K; q.bias = g.bias; q.M = M; q.N = N; q.K = K; q.ws = small_ws_;
    q.resid = g.resid; q.ldr = N;
    const bool need_x = g.out_x || (g.ln_g && K <= 576);
    if (need_x) { q.out_f32 = (float*)(base + oX); q.ldc32 = N; }
    if (K <= 576) {
      q.fsmn_v = g.fsmn_v; q.ldv = g.ldv; q.fsmn_wT = g.fsmn_wT; q.fsmn_k = g.fsmn_k; q.T = g.T;
      launch_gemm_small(stream_, q);
      if (g.ln_g) launch_layernorm(stream_, q.out_f32, M, N, g.ln_g, g.ln_b, g.out_n16, N, g.out_n32, N);
    } else {
      PF_CHECK(!g.fsmn_v, PF_ERR_INVALID_ARG, "gemm_rc: the split short-input form has no FSMN term");
      q.post_ln_g = g.ln_g; q.post_ln_b = g.ln_b; q.post_n16 = g.out_n16; q.ldn16 = N; q.post_n32 = g.out_n32; q.ldn32 = N;
      launch_gemm_small(stream_, q);
    }
  } else {
    prof_begin("gemm_op", 2.0 * M * (double)N * K);
    launch_gemm_rc(stream_, g);
    prof_end("gemm_op");
  }
  if (x_out) PF_HIP(hipMemcpyAsync(x_out, base + oX, (size_t)M * N * 4, hipMemcpyDeviceToHost, stream_));
  if (g.out_n32) PF_HIP(hipMemcpyAsync(n32_out, base + oN32, (size_t)M * N * 4, hipMemcpyDeviceToHost, stream_));
  std::vector<half_t> tmp;
  if (g.out_n16) {
    tmp.resize((size_t)M * N);
    PF_HIP(hipMemcpyAsync(tmp.data(), base + oN16, tmp.size() * 2, hipMemcpyDeviceToHost, stream_));
  }
  PF_HIP(hipStreamSynchronize(stream_));
  for (size_t i = 0; i < tmp.size(); ++i) n16_out[i] = (float)tmp[i];
}

// Encoder FFN as enc_layer() runs it: FFN-up writes the hidden in the blocked activation layout (kind 3),
// FFN-down reads it as a blocked A operand and adds bias + residual (kind 2).  y = resid + W2 relu(W1 x + b1) + b2.
void Engine::op_ffn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* resid, int M, int D, int F, float* y) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(M > 0 && D % 64 == 0 && F % 64 == 0, PF_ERR_INVALID_ARG, "ffn: D and F must be multiples of 64");
  const int64_t Mp = round_up(M, 256) + 128, Fp = round_up(F, 256), Dp = round_up(D, 256);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o32 = carve((size_t)std::max<int64_t>((int64_t)M * D, (int64_t)F * D) * 4);
  const size_t ox16 = carve((size_t)Mp * D * 2), ow1 = carve((size_t)Fp * D * 2), ow2 = carve((size_t)Dp * F * 2);
  const size_t ob1 = carve((size_t)F * 4), ob2 = carve((size_t)D * 4), oh = carve((size_t)Mp * F * 2);
  const size_t oxr = carve((size_t)Mp * D * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + ox16, 0, oxr - ox16, stream_));
  auto up16 = [&](const float* src, int rows, int cols, size_t dst) {
    PF_HIP(hipMemcpyAsync(base + o32, src, (size_t)rows * cols * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + o32), rows, cols, cols, (half_t*)(base + dst), cols);
    PF_HIP(hipStreamSynchronize(stream_));
  };
  up16(x, M, D, ox16); up16(w1, F, D, ow1); up16(w2, D, F, ow2);
  PF_HIP(hipMemcpyAsync(base + ob1, b1, (size_t)F * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ob2, b2, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + oxr, resid, (size_t)M * D * 4, hipMemcpyHostToDevice, stream_));
  Lin L1, L2;
  L1.w = (half_t*)(base + ow1); L1.bias = (const float*)(base + ob1); L1.N = F; L1.K = D; L1.Kpad = D;
  L2.w = (half_t*)(base + ow2); L2.bias = (const float*)(base + ob2); L2.N = D; L2.K = F; L2.Kpad = F;
  float* xr = (float*)(base + oxr);
  gemm("gemm_ffn1", L1, (half_t*)(base + ox16), D, M, nullptr, 0, (half_t*)(base + oh), F, nullptr, 0, nullptr, 0, true, 0, 1.f, true, 1);
  gemm("gemm_ffn2", L2, (half_t*)(base + oh), F, M, xr, D, nullptr, 0, xr, D, nullptr, 0, false, 0, 1.f, true, 2);
  PF_HIP(hipMemcpyAsync(y, xr, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

// The encoder FFN block as enc_layer() launches it for long inputs: retile W1 / W2, then ONE launch of ffn_fused_kernel.
// The decoder's FFN block in the split form of the fused kernel (k_ffn.hip), standalone: t = LN_F(relu(f16(x) W1^T + b1)) W2^T,
// n = LayerNorm(t); x = the block's normalised input, or (ds.ctx) LayerNorm norm1 of x_out = resid + ctx Wo^T + bo computed by
// the same launch.
void Engine::op_dec_ffn_fused(const pf_dec_ffn_desc& ds, float* t_out, float* n_out, float* x_out) {
  PF_HIP(hipSetDevice(device_));
  const int D = 512, F = 2048, M = ds.M, splits = ds.splits;
  PF_CHECK(M > 0, PF_ERR_INVALID_ARG, "dec_ffn_fused: M must be positive");
  PF_CHECK(splits == 0 || splits == 1 || splits == 2 || splits == 3 || splits == 4 || splits == 8, PF_ERR_INVALID_ARG,
           "dec_ffn_fused: splits must be 0 (automatic) | 1 | 2 | 3 | 4 | 8");
  const bool op = ds.ctx != nullptr;
  const int64_t Mp = round_up(M, 256) + 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o32 = carve((size_t)std::max<int64_t>((int64_t)M * D, (int64_t)F * D) * 4);
  const size_t ox16 = carve((size_t)Mp * D * 2), ow1 = carve((size_t)F * D * 2), ow2 = carve((size_t)D * F * 4);
  const size_t oimg = carve(ffn_dec_image_bytes()), ows = carve(ffn_dec_workspace_bytes(M, splits));
  const size_t ob1 = carve((size_t)F * 4), ogf = carve((size_t)F * 4), obf = carve((size_t)F * 4), og = carve((size_t)D * 4), obe = carve((size_t)D * 4);
  const size_t ot = carve((size_t)M * D * 4), on = carve((size_t)M * D * 4);
  const size_t owo = carve((size_t)D * D * 2), owot = carve(ffn_outproj_weight_bytes()), obo = carve((size_t)D * 4);
  const size_t og1 = carve((size_t)D * 4), obe1 = carve((size_t)D * 4), oxr = carve((size_t)Mp * D * 4), oxo = carve((size_t)Mp * D * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + ox16, 0, (size_t)Mp * D * 2, stream_));
  auto up16 = [&](const float* src, int rows, int cols, size_t dst, int ldo) {
    PF_HIP(hipMemcpyAsync(base + o32, src, (size_t)rows * cols * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + o32), rows, cols, cols, (half_t*)(base + dst), ldo);
    PF_HIP(hipStreamSynchronize(stream_));
  };
  auto up = [&](const float* src, size_t n, size_t dst) { PF_HIP(hipMemcpyAsync(base + dst, src, n * 4, hipMemcpyHostToDevice, stream_)); };
  up16(op ? ds.ctx : ds.x, M, D, ox16, D); up16(ds.w1, F, D, ow1, D);
  up(ds.w2, (size_t)D * F, ow2); up(ds.b1, F, ob1); up(ds.gamma_f, F, ogf); up(ds.beta_f, F, obf);
  if (ds.ln_gamma) { up(ds.ln_gamma, D, og); up(ds.ln_beta, D, obe); }
  launch_ffn_dec_retile(stream_, (const half_t*)(base + ow1), D, (const float*)(base + ow2), (const float*)(base + ogf),
                        (const float*)(base + obf), (const float*)(base + ob1), (half_t*)(base + oimg));
  FfnDecArgs f{};
  f.A = (const half_t*)(base + ox16); f.lda = D; f.img = (const half_t*)(base + oimg); f.ws = base + ows; f.M = M; f.splits = splits;
  f.eps_hidden = 1e-12f; f.eps = 1e-12f;
  if (op) {
    up16(ds.wo, D, D, owo, D);
    launch_ffn_retile_out(stream_, (const half_t*)(base + owo), D, (half_t*)(base + owot));
    up(ds.bo, D, obo); up(ds.ln1_gamma, D, og1); up(ds.ln1_beta, D, obe1);
    PF_HIP(hipMemsetAsync(base + oxr, 0, (size_t)Mp * D * 4, stream_));
    up(ds.resid, (size_t)M * D, oxr);
    f.A = nullptr; f.ctx = (const half_t*)(base + ox16); f.lda_c = D; f.Wot = (const half_t*)(base + owot); f.bo = (const float*)(base + obo);
    f.resid = (const float*)(base + oxr); f.ldr = D; f.out_x = (float*)(base + oxo); f.ldx = D;
    f.ln1_g = (const float*)(base + og1); f.ln1_b = (const float*)(base + obe1); f.eps1 = 1e-12f;
  }
  if (t_out) { f.t32 = (float*)(base + ot); f.ldt = D; }
  if (ds.ln_gamma) { f.ln_g = (const float*)(base + og); f.ln_b = (const float*)(base + obe); }
  if (n_out) { f.n32 = (float*)(base + on); f.ldn32 = D; }
  const char* rep = getenv("PF_OP_REPEAT");                        // tools/: repeated launches, timed as class "gemm_op_warm"
  const int reps = rep ? std::max(1, atoi(rep)) : 1;
  for (int r = 0; r < reps; ++r) {
    prof_begin(r == 0 ? "gemm_op" : "gemm_op_warm", 4.0 * M * (double)D * F + (op ? 2.0 * M * (double)D * D : 0.0));
    launch_ffn_dec(stream_, f);
    prof_end(r == 0 ? "gemm_op" : "gemm_op_warm");
  }
  if (t_out) PF_HIP(hipMemcpyAsync(t_out, base + ot, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  if (n_out) PF_HIP(hipMemcpyAsync(n_out, base + on, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  if (x_out && op) PF_HIP(hipMemcpyAsync(x_out, base + oxo, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_ffn_fused(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* resid,
                          const float* g, const float* be, int M, float* x_out, float* n16_out, const pf_attn_ffn_desc* op) {
  PF_HIP(hipSetDevice(device_));
  const int D = 512, F = 2048;
  PF_CHECK(M > 0, PF_ERR_INVALID_ARG, "ffn_fused: M must be positive");
  PF_CHECK(op || x, PF_ERR_INVALID_ARG, "ffn_fused: missing operand");
  const int64_t Mp = round_up(M, 256) + 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o32 = carve((size_t)std::max<int64_t>((int64_t)M * D, (int64_t)F * D) * 4);
  const size_t ox16 = carve((size_t)Mp * D * 2), ow1 = carve((size_t)F * D * 2), ow2 = carve((size_t)D * F * 2);
  const size_t owt = carve(ffn_fused_weight_bytes());
  const size_t ob1 = carve((size_t)F * 4), ob2 = carve((size_t)D * 4), og = carve((size_t)D * 4), obe = carve((size_t)D * 4);
  const size_t oxr = carve((size_t)Mp * D * 4), oxo = carve((size_t)Mp * D * 4), on16 = carve((size_t)Mp * D * 2);
  // out-projection form: Wo (f16 + its image), bias, the V slice inside a [M, 3 D] QKV-shaped buffer, taps, norm2, x_mid scratch
  const size_t owo = carve((size_t)D * D * 2), owot = carve(ffn_outproj_weight_bytes()), obo = carve((size_t)D * 4);
  const size_t ov = carve((size_t)(Mp + 128) * 3 * D * 2), owT = carve((size_t)11 * D * 4), og2 = carve((size_t)D * 4), obe2 = carve((size_t)D * 4);
  const bool tail = op && op->wqkv;
  const size_t owq = carve((size_t)3 * D * D * 2), owqt = carve(3 * ffn_outproj_weight_bytes()), obq = carve((size_t)3 * D * 4);
  const size_t oqk = carve((size_t)Mp * 2 * D * 2), ovo = carve((size_t)Mp * D * 2);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + ox16, 0, (size_t)Mp * D * 2, stream_));
  PF_HIP(hipMemsetAsync(base + oxr, 0, (size_t)Mp * D * 4, stream_));
  auto up16 = [&](const float* src, int rows, int cols, size_t dst, int ldo) {
    PF_HIP(hipMemcpyAsync(base + o32, src, (size_t)rows * cols * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + o32), rows, cols, cols, (half_t*)(base + dst), ldo);
    PF_HIP(hipStreamSynchronize(stream_));
  };
  up16(op ? op->ctx : x, M, D, ox16, D); up16(w1, F, D, ow1, D); up16(w2, D, F, ow2, F);
  PF_HIP(hipMemcpyAsync(base + ob1, b1, (size_t)F * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ob2, b2, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
  if (resid) PF_HIP(hipMemcpyAsync(base + oxr, resid, (size_t)M * D * 4, hipMemcpyHostToDevice, stream_));
  launch_ffn_retile(stream_, (half_t*)(base + ow1), D, (half_t*)(base + ow2), F, (half_t*)(base + owt));
  FfnFusedArgs f{};
  f.A = (half_t*)(base + ox16); f.lda = D; f.Wt = (half_t*)(base + owt); f.b1 = (const float*)(base + ob1); f.b2 = (const float*)(base + ob2);
  f.M = M; f.resid = resid || !op ? (const float*)(base + oxr) : nullptr; f.ldr = D; f.eps = 1e-12f;
  std::vector<float> wT;
  if (op) {
    up16(op->wo, D, D, owo, D);
    launch_ffn_retile_out(stream_, (half_t*)(base + owo), D, (half_t*)(base + owot));
    PF_HIP(hipMemsetAsync(base + ov, 0, (size_t)(Mp + 128) * 3 * D * 2, stream_));
    PF_HIP(hipMemcpyAsync(base + o32, op->v, (size_t)M * D * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + o32), M, D, D, (half_t*)(base + ov) + 2 * D, 3 * D);
    PF_HIP(hipStreamSynchronize(stream_));
    wT.resize((size_t)11 * D);
    for (int c = 0; c < D; ++c)
      for (int j = 0; j < 11; ++j) wT[(size_t)j * D + c] = op->fsmn_w[(size_t)c * 11 + j];
    PF_HIP(hipMemcpyAsync(base + owT, wT.data(), wT.size() * 4, hipMemcpyHostToDevice, stream_));
    PF_HIP(hipMemcpyAsync(base + obo, op->bo, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
    PF_HIP(hipMemcpyAsync(base + og2, op->ln2_gamma, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
    PF_HIP(hipMemcpyAsync(base + obe2, op->ln2_beta, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
    f.ctx = (half_t*)(base + ox16); f.lda_c = D; f.Wot = (half_t*)(base + owot); f.bo = (const float*)(base + obo);
    f.fsmn_v = (half_t*)(base + ov) + 2 * D; f.ldv = 3 * D; f.fsmn_wT = (const float*)(base + owT); f.T = op->T > 0 ? op->T : M;
    f.ln2_g = (const float*)(base + og2); f.ln2_b = (const float*)(base + obe2);
    f.A = nullptr;
  }
  if (tail) {
    PF_CHECK(op->bqkv && g, PF_ERR_INVALID_ARG, "attn_ffn_fused: the Q | K | V tail needs its bias and the LayerNorm in front of it");
    up16(op->wqkv, 3 * D, D, owq, D);
    for (int part = 0; part < 3; ++part)
      launch_ffn_retile_out(stream_, (half_t*)(base + owq) + (size_t)part * D * D, D,
                            (half_t*)(base + owqt) + (size_t)part * (ffn_outproj_weight_bytes() / 2));
    PF_HIP(hipMemcpyAsync(base + obq, op->bqkv, (size_t)3 * D * 4, hipMemcpyHostToDevice, stream_));
    f.Wqt = (half_t*)(base + owqt); f.bq = (const float*)(base + obq); f.out_qk = (half_t*)(base + oqk); f.out_v = (half_t*)(base + ovo);
    f.ldvo = D; f.qscale = 1.0f / std::sqrt(128.0f);
  }
  if (x_out) { f.out_x = (float*)(base + oxo); f.ldx = D; }
  if (g) {
    PF_HIP(hipMemcpyAsync(base + og, g, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
    PF_HIP(hipMemcpyAsync(base + obe, be, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
    f.ln_g = (const float*)(base + og); f.ln_b = (const float*)(base + obe);
    if (n16_out) { f.out_n16 = (half_t*)(base + on16); f.ldn16 = D; }
  }
  const char* rep = getenv("PF_OP_REPEAT");                        // tools/: repeated launches, timed as class "gemm_op_warm"
  const int reps = rep ? std::max(1, atoi(rep)) : 1;
  for (int r = 0; r < reps; ++r) {                                 // (resid is a separate buffer: repeats compute the same result)
    prof_begin(r == 0 ? "gemm_op" : "gemm_op_warm", 4.0 * M * (double)D * F + (op ? 2.0 * M * (double)D * D : 0.0));
    launch_ffn_fused(stream_, f);
    prof_end(r == 0 ? "gemm_op" : "gemm_op_warm");
  }
  if (x_out) PF_HIP(hipMemcpyAsync(x_out, base + oxo, (size_t)M * D * 4, hipMemcpyDeviceToHost, stream_));
  std::vector<half_t> n16;
  if (f.out_n16) {
    n16.resize((size_t)M * D);
    PF_HIP(hipMemcpyAsync(n16.data(), base + on16, n16.size() * 2, hipMemcpyDeviceToHost, stream_));
  }
  PF_HIP(hipStreamSynchronize(stream_));
  if (f.out_n16)
    for (size_t i = 0; i < n16.size(); ++i) n16_out[i] = (float)n16[i];
  if (tail) {
    std::vector<half_t> qk((size_t)Mp * 2 * D), vv((size_t)M * D);
    PF_HIP(hipMemcpy(qk.data(), base + oqk, qk.size() * 2, hipMemcpyDeviceToHost));
    PF_HIP(hipMemcpy(vv.data(), base + ovo, vv.size() * 2, hipMemcpyDeviceToHost));
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < 2 * D; ++n) {                       // blocked [Mpad, 1024]: ((m / 32 * 128 + n / 8) * 32 + m % 32) * 8 + n % 8
        const float val = (float)qk[(((size_t)(m >> 5) * 128 + (n >> 3)) * 32 + (m & 31)) * 8 + (n & 7)];
        float* dst = n < D ? op->q_out : op->k_out;
        if (dst) dst[(size_t)m * D + (n & (D - 1))] = val;
      }
    if (op->v_out)
      for (size_t i = 0; i < vv.size(); ++i) op->v_out[i] = (float)vv[i];
  }
}

// Encoder FSMN exactly as enc_layer() launches it: the f16 V slice of a [M, 3D] QKV buffer (row stride 3D).
void Engine::op_fsmn_enc(const float* v, const float* w, int B, int T, int D, int k, float* y) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(D % 8 == 0 && B > 0 && T > 0, PF_ERR_INVALID_ARG, "fsmn_enc: bad shape");
  const size_t n = (size_t)B * T * D;
  std::vector<float> wT((size_t)D * k);
  for (int c = 0; c < D; ++c)
    for (int j = 0; j < k; ++j) wT[(size_t)j * D + c] = w[(size_t)c * k + j];
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o2 = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o2; };
  const size_t ov = carve(n * 4), oq = carve(((size_t)B * T + 128) * 3 * D * 2), ow = carve(wT.size() * 4), oy = carve(n * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + oq, 0, ((size_t)B * T + 128) * 3 * D * 2, stream_));
  PF_HIP(hipMemcpyAsync(base + ov, v, n * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ow, wT.data(), wT.size() * 4, hipMemcpyHostToDevice, stream_));
  half_t* vs = (half_t*)(base + oq) + 2 * D;
  launch_f32_to_f16(stream_, (const float*)(base + ov), (int64_t)B * T, D, D, vs, 3 * D);
  launch_fsmn_enc(stream_, vs, 3 * D, (const float*)(base + ow), B, T, D, k, (float*)(base + oy));
  PF_HIP(hipMemcpyAsync(y, base + oy, n * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

// Decoder FSMN exactly as the decoder launches it: x += (dwconv(tn*m) + tn*m)*m, m = (l < token_num[b]).
void Engine::op_fsmn_dec(const float* tn, const float* w, const int32_t* token_num, int B, int L, int D, int k, float* x) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(D % 4 == 0 && B > 0 && L > 0, PF_ERR_INVALID_ARG, "fsmn_dec: bad shape");
  const size_t n = (size_t)B * L * D;
  std::vector<float> wT((size_t)D * k);
  for (int c = 0; c < D; ++c)
    for (int j = 0; j < k; ++j) wT[(size_t)j * D + c] = w[(size_t)c * k + j];
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o2 = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o2; };
  const size_t ot = carve(n * 4), ox = carve(n * 4), ow = carve(wT.size() * 4), on = carve((size_t)B * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemcpyAsync(base + ot, tn, n * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ox, x, n * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ow, wT.data(), wT.size() * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + on, token_num, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
  launch_fsmn_dec(stream_, (const float*)(base + ot), (const float*)(base + ow), (const int32_t*)(base + on), B, L, D, k,
                  (float*)(base + ox));
  PF_HIP(hipMemcpyAsync(x, base + ox, n * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

// The pipeline's vocabulary tail: log-probs y = (x - max) - log(sum exp(x - max)) and the reference's last-index
// arg-max over y (OfflineRecognizer.cs:139-152 scans the graph OUTPUT).  y == nullptr: ids only (mode 1).
void Engine::op_logsoftmax_argmax(const float* x, int64_t rows, int V, float* y, int64_t* ids) {
  PF_HIP(hipSetDevice(device_));
  if (rows == 0) return;
  const int ld = (int)round_up(V, 4);
  ensure(ws_tmp_, (size_t)rows * ld * 4 + (size_t)rows * 8 + 256);
  float* xd = (float*)ws_tmp_.p;
  int64_t* idd = (int64_t*)((char*)ws_tmp_.p + round_up((int64_t)rows * ld * 4, 256));
  PF_HIP(hipMemcpy2DAsync(xd, (size_t)ld * 4, x, (size_t)V * 4, (size_t)V * 4, rows, hipMemcpyHostToDevice, stream_));
  launch_argmax(stream_, xd, rows, V, ld, y ? 2 : 1, idd);
  if (y) PF_HIP(hipMemcpy2DAsync(y, (size_t)V * 4, xd, (size_t)ld * 4, (size_t)V * 4, rows, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(ids, idd, (size_t)rows * 8, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_layernorm(const float* x, const float* g, const float* b, int64_t rows, int D, float* y) {
  PF_HIP(hipSetDevice(device_));
  if (rows == 0) return;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t ox = carve((size_t)rows * D * 4), og = carve((size_t)D * 4), obb = carve((size_t)D * 4), oy = carve((size_t)rows * D * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemcpyAsync(base + ox, x, (size_t)rows * D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + og, g, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + obb, b, (size_t)D * 4, hipMemcpyHostToDevice, stream_));
  launch_layernorm(stream_, (const float*)(base + ox), rows, D, (const float*)(base + og), (const float*)(base + obb),
                   nullptr, 0, (float*)(base + oy), D);
  PF_HIP(hipMemcpyAsync(y, base + oy, (size_t)rows * D * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_attention(const float* q, const float* k, const float* v, int B, int Lq, int Lk, int H, float* o) {
  PF_HIP(hipSetDevice(device_));
  const int Dm = H * 128;
  const int64_t nq = (int64_t)B * Lq, nk = (int64_t)B * Lk;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o2 = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o2; };
  const size_t oin = carve((size_t)std::max(nq, nk) * Dm * 4);
  const size_t oq = carve((size_t)(nq + 128) * Dm * 2), ok = carve((size_t)(nk + 128) * Dm * 2);
  const size_t ov = carve((size_t)(nk + 128) * Dm * 2), oo = carve((size_t)(nq + 128) * Dm * 2), oo32 = carve((size_t)nq * Dm * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + oq, 0, off - oq, stream_));
  auto up = [&](const float* src, int64_t rows, size_t dst) {
    PF_HIP(hipMemcpyAsync(base + oin, src, (size_t)rows * Dm * 4, hipMemcpyHostToDevice, stream_));
    launch_f32_to_f16(stream_, (const float*)(base + oin), rows, Dm, Dm, (half_t*)(base + dst), Dm);
    PF_HIP(hipStreamSynchronize(stream_));
  };
  up(q, nq, oq); up(k, nk, ok); up(v, nk, ov);
  AttnArgs a{};
  a.q = (half_t*)(base + oq); a.k = (half_t*)(base + ok); a.v = (half_t*)(base + ov); a.o = (half_t*)(base + oo);
  a.q_bstride = (int64_t)Lq * Dm; a.k_bstride = a.v_bstride = (int64_t)Lk * Dm; a.o_bstride = (int64_t)Lq * Dm;
  a.q_rstride = a.k_rstride = a.v_rstride = a.o_rstride = Dm;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  prof_begin("attn_op", 4.0 * B * (double)Lq * Lk * Dm);
  launch_attention(stream_, a);
  prof_end("attn_op");
  // f16 -> f32 on the host side of the copy
  std::vector<uint16_t> tmp((size_t)nq * Dm);
  PF_HIP(hipMemcpyAsync(tmp.data(), base + oo, tmp.size() * 2, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  for (size_t i = 0; i < tmp.size(); ++i) {
    half_t hv;
    std::memcpy(&hv, &tmp[i], 2);
    o[i] = (float)hv;
  }
  (void)oo32;
}

// The encoder's fused Q | K | V projection and its self-attention as enc_layer() launches them for long inputs: the persistent
// 256 x 192 kernel (Q scaled and K blocked, V row-major: k_gemm_qkv.hip) followed by the attention kernel reading that layout.
// x [B*T, K], w [1536, K] ([Q | K | V] rows), bias [1536] or null; outputs (each may be null) q / k / v / ctx [B*T, 512] as
// fp32 copies of the stored f16 values (q / k de-blocked on the host).
void Engine::op_qkv_attention(const float* x, const float* w, const float* bias, int B, int T, int K, float* q_out, float* k_out,
                              float* v_out, float* ctx_out) {
  PF_HIP(hipSetDevice(device_));
  const int D = 512, H = 4, N = 3 * D;
  const int M = B * T;
  PF_CHECK(M > 0 && K > 0, PF_ERR_INVALID_ARG, "qkv_attention: empty input");
  const int Kpad = (int)round_up(K, 64);
  PF_CHECK(gemm_qkvp_applicable(M, Kpad, Kpad, Kpad, D), PF_ERR_INVALID_ARG, "qkv_attention: shape not covered by the 256 x 192 kernel");
  const int64_t Mp = round_up(M, 256) + 128;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o2 = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o2; };
  const size_t o32 = carve((size_t)std::max<int64_t>((int64_t)M * K, (int64_t)N * K) * 4), oA = carve((size_t)Mp * Kpad * 2);
  const size_t oW = carve((size_t)N * Kpad * 2), oWp = carve((size_t)N * Kpad * 2), ob = carve((size_t)N * 4), obp = carve((size_t)N * 4);
  const size_t oqk = carve((size_t)Mp * 2 * D * 2), ov = carve((size_t)Mp * D * 2), oc = carve((size_t)Mp * D * 2);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemsetAsync(base + oA, 0, (size_t)Mp * Kpad * 2, stream_));
  PF_HIP(hipMemsetAsync(base + oW, 0, (size_t)N * Kpad * 2, stream_));
  PF_HIP(hipMemcpyAsync(base + o32, x, (size_t)M * K * 4, hipMemcpyHostToDevice, stream_));
  launch_f32_to_f16(stream_, (const float*)(base + o32), M, K, K, (half_t*)(base + oA), Kpad);
  PF_HIP(hipStreamSynchronize(stream_));
  PF_HIP(hipMemcpyAsync(base + o32, w, (size_t)N * K * 4, hipMemcpyHostToDevice, stream_));
  launch_f32_to_f16(stream_, (const float*)(base + o32), N, K, K, (half_t*)(base + oW), Kpad);
  if (bias) PF_HIP(hipMemcpyAsync(base + ob, bias, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
  launch_qkv_permute(stream_, (half_t*)(base + oW), Kpad, bias ? (const float*)(base + ob) : nullptr, (half_t*)(base + oWp), (float*)(base + obp));
  const float qscale = 1.0f / std::sqrt((float)(D / H));
  half_t* qk = (half_t*)(base + oqk);
  half_t* vb = (half_t*)(base + ov);
  prof_begin("gemm_op", 2.0 * M * (double)N * K);
  launch_gemm_qkvp(stream_, (half_t*)(base + oA), Kpad, (half_t*)(base + oWp), Kpad, (const float*)(base + obp), M, Kpad, qscale, qk, vb, D);
  prof_end("gemm_op");
  AttnArgs a{};
  a.q = qk; a.k = qk; a.qk_blocked = 1; a.blk_groups = 2 * D / 8; a.blk_brows = T; a.blk_kgrp = D / 8;
  a.v = vb; a.v_bstride = (int64_t)T * D; a.v_rstride = D;
  a.o = (half_t*)(base + oc); a.o_bstride = (int64_t)T * D; a.o_rstride = D;
  a.q_rstride = a.k_rstride = 8;                          // ignored (alignment checks only)
  a.B = B; a.H = H; a.Lq = T; a.Lk = T;
  prof_begin("attn_op", 4.0 * B * (double)T * T * D);
  launch_attention(stream_, a);
  prof_end("attn_op");
  std::vector<half_t> hqk((size_t)Mp * 2 * D), hv((size_t)M * D), hc((size_t)M * D);
  PF_HIP(hipMemcpyAsync(hqk.data(), qk, hqk.size() * 2, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(hv.data(), vb, hv.size() * 2, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(hc.data(), base + oc, hc.size() * 2, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  const int G = 2 * D / 8;
  for (int m = 0; m < M; ++m)
    for (int c = 0; c < 2 * D; ++c) {
      const float val = (float)hqk[(((size_t)(m >> 5) * G + (c >> 3)) * 32 + (m & 31)) * 8 + (c & 7)];
      if (c < D) { if (q_out) q_out[(size_t)m * D + c] = val; }
      else if (k_out) k_out[(size_t)m * D + c - D] = val;
    }
  for (size_t i = 0; i < hv.size(); ++i) {
    if (v_out) v_out[i] = (float)hv[i];
    if (ctx_out) ctx_out[i] = (float)hc[i];
  }
}

void Engine::op_fsmn(const float* v, const float* w, const float* mask, int B, int T, int D, int k, float* y) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(D % 4 == 0, PF_ERR_INVALID_ARG, "fsmn: D must be a multiple of 4");
  const size_t n = (size_t)B * T * D;
  if (n == 0) return;
  std::vector<float> wT((size_t)D * k);
  for (int c = 0; c < D; ++c)
    for (int j = 0; j < k; ++j) wT[(size_t)j * D + c] = w[(size_t)c * k + j];
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o2 = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o2; };
  const size_t ov = carve(n * 4), ow = carve(wT.size() * 4), om = carve((size_t)B * T * 4), oy = carve(n * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemcpyAsync(base + ov, v, n * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + ow, wT.data(), wT.size() * 4, hipMemcpyHostToDevice, stream_));
  if (mask) PF_HIP(hipMemcpyAsync(base + om, mask, (size_t)B * T * 4, hipMemcpyHostToDevice, stream_));
  launch_fsmn_f32(stream_, (const float*)(base + ov), (const float*)(base + ow), mask ? (const float*)(base + om) : nullptr,
                  B, T, D, k, (float*)(base + oy));
  PF_HIP(hipMemcpyAsync(y, base + oy, n * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_cif(const float* H, const float* alphas, int B, int T, int D, float thr, int Lcap, float* E,
                    int32_t* fire_count, int32_t* token_num, int32_t* L_out) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(D % 4 == 0 && B > 0 && T > 0, PF_ERR_INVALID_ARG, "cif: bad shape");
  const int T1 = T + 1;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o2 = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o2; };
  const size_t oH = carve((size_t)B * T * D * 4), oa = carve((size_t)B * T1 * 4), ofc = carve((size_t)B * 4), otn = carve((size_t)B * 4);
  const size_t off_ = carve((size_t)B * T1 * 4), owc = carve((size_t)B * T1 * 4), owr = carve((size_t)B * T1 * 4), omx = carve(256);
  const size_t oE = carve((size_t)B * std::max(Lcap, 1) * D * 4);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  CifPlan p;
  p.fire_count = (int32_t*)(base + ofc); p.token_num = (int32_t*)(base + otn); p.fire_frame = (int32_t*)(base + off_);
  p.w_cur = (float*)(base + owc); p.w_rem = (float*)(base + owr); p.max_count = (int32_t*)(base + omx);
  PF_HIP(hipMemcpyAsync(base + oH, H, (size_t)B * T * D * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + oa, alphas, (size_t)B * T1 * 4, hipMemcpyHostToDevice, stream_));
  if (mc_.cif_cumsum) launch_cif_scan_cumsum(stream_, (const float*)(base + oa), B, T1, p);
  else launch_cif_scan(stream_, (const float*)(base + oa), B, T1, thr, p);
  int32_t L = 0;
  PF_HIP(hipMemcpyAsync(&L, p.max_count, 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(fire_count, p.fire_count, (size_t)B * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipMemcpyAsync(token_num, p.token_num, (size_t)B * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  if (L_out) *L_out = L;
  PF_CHECK(L <= Lcap, PF_ERR_CAPACITY, "cif: Lcap " + std::to_string(Lcap) + " < L = " + std::to_string(L));
  if (Lcap == 0) return;
  if (mc_.cif_cumsum) launch_cif_gather_cumsum(stream_, (const float*)(base + oH), (const float*)(base + oa), B, T, D, T1, p, Lcap, (float*)(base + oE));
  else launch_cif_gather(stream_, (const float*)(base + oH), B, T, D, T1, p, Lcap, (float*)(base + oE));
  PF_HIP(hipMemcpyAsync(E, base + oE, (size_t)B * Lcap * D * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

void Engine::op_encoder(const float* speech, int B, int T, float* H) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(speech && H && B > 0 && T > 0, PF_ERR_INVALID_ARG, "encoder: bad arguments");
  const size_t n = (size_t)B * T * mc_.feat_dim;
  ensure(ws_speech_, n * 4);
  PF_HIP(hipMemcpyAsync(ws_speech_.p, speech, n * 4, hipMemcpyHostToDevice, stream_));
  encoder((const float*)ws_speech_.p, B, T);
  PF_HIP(hipMemcpyAsync(H, H32_, (size_t)B * T * mc_.d_model * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
}

}  // namespace pf
