// engine_fp32.cpp — the fp32 graph of the engine (math_mode 1: fp32 MFMA parity mode; math_mode 3: "exact" at matrix-core
// speed, every large Linear as three f16 MFMA products of (hi, lo) operand pairs — DESIGN.md section 3), incl. the fp32 forms of
// the BiCIF timestamp head and the SeACo bias branch.  Split out of engine.cpp in round 6; the nodes are those
// behind AliParaformerAsr/OfflineProjOfParaformer.cs:68 / OfflineProjOfSeacoParaformer.cs:48-135.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>

#include "hostutil.h"

namespace pf {

static const size_t kAlign = 256;

// ------------------------------------------------------------------ fp32 parity mode -------
// math_mode = 1: the same graph with fp32 activations and fp32 weights on the fp32 matrix path (k_fp32.hip).  Supported
// for the paraformer and SenseVoice graphs (no BiCIF head, no SeACo branch); one launch per graph node, no fusion.
enum { F_X = 0, F_XN, F_Q, F_K, F_V, F_CTX, F_FS, F_H, F_T, F_COUNT };

// One Linear of the fp32 graph.  math_mode 1: exact fp32 products on v_mfma_f32_32x32x2_f32 (k_fp32.hip).  math_mode 3
// ("exact", round 5): operands as hi + 2^-11 lo' pairs of f16 numbers (22 mantissa bits, launch_split_x3) and
//   x W^T = hi_x hi_W^T + 2^-11 (hi_x lo'_W^T + lo'_x hi_W^T)            (the lo lo term is 2^-22 relative: dropped)
// as TWO launches of the pipeline's own f16 MFMA kernels with fp32 results: [hi_x] x [hi_W] (depth K) -> t, then
// [hi_x | lo'_x] x [lo'_W | hi_W] (depth 2 K) scaled by 2^-11 in the epilogue, + t, + residual, ReLU — three times the f16
// MFMA work at 16x the fp32 matrix rate.  The weight pairs are built on first use and kept (same bytes as the fp32 matrix).
// flags: kX3OutPair — the result is only the A operand of the NEXT gemm32 (FFN hidden): in mode 3 it is written as its
// (hi | lo') pair by the product's epilogue and `out` is not touched; kX3InPair — A is that pair (the `A` pointer is ignored in
// mode 3).  Mode 1 ignores both flags.  resid2: a second fp32 addend with the row stride of resid (the FSMN memory beside the
// residual stream).
// attention of the fp32 graph: math_mode 1 on the fp32 matrix path; math_mode 3: PF_X3_ATTN = 0 the same, 1 = x3 operands
// throughout, 2 = fp32 scores (what is exponentiated stays exact) + x3 operands for P V
void Engine::attention32(const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs, const float* v, int64_t v_bs,
                         int v_rs, float* o, int64_t o_bs, int o_rs, int B, int H, int Lq, int Lk, bool only_operand) {
  const char* acls = (q_rs == k_rs && Lq == Lk) ? "attn32_self" : "attn32_cross";
  prof_begin(acls, 4.0 * B * H * (double)Lq * Lk * 128);
  struct End { Engine* e; const char* c; ~End() { e->prof_end(c); } } end_{this, acls};
  // only_operand: o [B * Lq, H * 128] (dense rows) is nothing but the A operand of the gemm32 that follows — in math_mode 3 the
  // fp32-MFMA kernel's epilogue writes it as that product's (hi | lo') pair (no fp32 context, no split pass)
  const int Dm = H * 128, M = B * Lq;
  if (only_operand && x3_mode_ && x3_fuse_ && x3_attn_ == 0 && o_rs == Dm && o_bs == (int64_t)Lq * Dm && Dm % 64 == 0 && M > gemm_small_max_rows()) {
    const int64_t Mp = round_up(M, 256) + 128;
    ensure(ws_x3a_, (size_t)Mp * 2 * Dm * 2);
    half_t* a2 = (half_t*)ws_x3a_.p;
    if (launch_attention_f32_pair(stream_, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, a2, (int64_t)Lq * 2 * Dm, 2 * Dm, Dm, B, H, Lq, Lk)) {
      x3a_src_ = o; x3a_M_ = M; x3a_K_ = Dm; x3a_ld_ = Dm; x3a_buf_ = a2; x3a_pair_only_ = true;
      return;
    }
  }
  if (x3_mode_ && x3_attn_ == 1) launch_attention_x3(stream_, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk, false);
  else if (x3_mode_ && x3_attn_ == 2) launch_attention_x3(stream_, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk, true);
  else launch_attention_f32(stream_, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk);
}

void Engine::layernorm32(const float* x, int M, int D, const LNp& ln, float* xn) {
  prof_begin("layernorm", 0);
  struct End { Engine* e; ~End() { e->prof_end("layernorm"); } } end_{this};
  if (x3_mode_ && x3_fuse_ && D == 512 && M > gemm_small_max_rows()) {
    const int64_t Mp = round_up(M, 256) + 128;
    ensure(ws_x3a_, (size_t)Mp * 2 * D * 2);
    half_t* a2 = (half_t*)ws_x3a_.p;
    launch_layernorm_pair(stream_, x, M, ln.g, ln.b, a2, 2 * D, D);
    x3a_src_ = xn; x3a_M_ = M; x3a_K_ = D; x3a_ld_ = D; x3a_buf_ = a2; x3a_pair_only_ = true;
    return;
  }
  launch_layernorm(stream_, x, M, D, ln.g, ln.b, nullptr, 0, xn, D);
}

void Engine::gemm32(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K, float* out, int ldc,
                    const float* resid, int ldr, bool relu, int scale_cols, float scale, int flags, const float* resid2) {
  // the class's FLOPs are the fp32 graph's 2 M N K; math_mode 3 executes three times that on the f16 matrix cores
  const char* cls = cls32_;
  prof_begin(cls, 2.0 * M * (double)N * K);
  gemm32_impl(A, lda, W, ldw, bias, M, N, K, out, ldc, resid, ldr, relu, scale_cols, scale, flags, resid2);
  prof_end(cls);
}

void Engine::gemm32_impl(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K, float* out, int ldc,
                         const float* resid, int ldr, bool relu, int scale_cols, float scale, int flags, const float* resid2) {
  // (a q-scale on the leading columns only — the fused Q | K | V product — is an option of the one-launch form's epilogue)
  const bool part_scale = scale_cols > 0 && scale_cols < N;
  const bool x3 = x3_mode_ && M >= 64 && ldw == K && (!part_scale || (x3_one_ && scale_cols % 64 == 0 && M > gemm_small_max_rows())) &&
                  (ldc % 4) == 0 && (!resid || ldr % 4 == 0);
  const bool pair_ok = x3 && M > gemm_small_max_rows();
  if (!x3) {
    PF_CHECK(!(flags & kX3InPair) || !x3_pair_live_, PF_ERR_UNSUPPORTED, "gemm32: operand pair without its consumer");
    PF_CHECK(!(x3a_pair_only_ && x3a_src_ == A), PF_ERR_UNSUPPORTED, "gemm32: the operand exists only as an x3 pair, but this product does not take the x3 form");
    launch_gemm_f32(stream_, A, lda, W, ldw, bias, M, N, K, out, ldc, resid, ldr, relu, scale_cols, scale);
    if (resid2) {
      PF_CHECK(ldc == N && ldr == N, PF_ERR_UNSUPPORTED, "gemm32: a second addend needs contiguous rows");
      launch_add_f32(stream_, out, resid2, (int64_t)M * N);
    }
    return;
  }
  const int Kp = (int)round_up(K, 64);
  const int64_t Np = round_up(N, 256), Mp = round_up(M, 256) + 128;
  auto it = x3w_.find(std::make_pair(W, N));
  if (it == x3w_.end()) {
    half_t* wc = (half_t*)dalloc((size_t)Np * 2 * Kp * 2);
    PF_HIP(hipMemsetAsync(wc, 0, (size_t)Np * 2 * Kp * 2, stream_));
    launch_split_x3(stream_, W, N, K, ldw, wc, 2 * Kp, Kp, 1);        // rows = [lo'_W | hi_W]
    it = x3w_.emplace(std::make_pair(W, N), wc).first;
  }
  const half_t* wcat = it->second;
  const int ldt = (int)round_up(N, 4);
  ensure(ws_x3t_, (size_t)Mp * ldt * 4);
  float* t = (float*)ws_x3t_.p;
  half_t* a2;
  if ((flags & kX3InPair) && x3_pair_live_) {
    PF_CHECK(x3_pair_M_ == M && x3_pair_K_ == K, PF_ERR_INVALID_ARG, "gemm32: operand pair of another shape");
    a2 = (half_t*)ws_x3h_.p;                                          // written by the producing product's epilogue
    x3_pair_live_ = false; x3a_src_ = nullptr; x3a_pair_only_ = false;
  } else {
    ensure(ws_x3a_, (size_t)Mp * 2 * Kp * 2);
    a2 = (half_t*)ws_x3a_.p;
    // kX3SameInput: the caller states that A is the (unchanged) operand of its previous gemm32 call — Q, K and V share one
    // (or the producer wrote the pair itself: layernorm32 / attention32 — then there is no fp32 form to split)
    const bool same = ((flags & kX3SameInput) || x3a_pair_only_) && x3a_src_ == A && x3a_M_ == M && x3a_K_ == K && x3a_ld_ == lda && x3a_buf_ == a2;
    PF_CHECK(same || !(x3a_pair_only_ && x3a_src_ == A), PF_ERR_UNSUPPORTED, "gemm32: the operand pair in the arena is not the one this product names");
    if (!same) { launch_split_x3(stream_, A, M, K, lda, a2, 2 * Kp, Kp, 0); x3a_pair_only_ = false; }   // rows = [hi_x | lo'_x]
    x3a_src_ = A; x3a_M_ = M; x3a_K_ = K; x3a_ld_ = lda; x3a_buf_ = a2;
  }
  const bool out_pair = (flags & kX3OutPair) && pair_ok && N % 64 == 0;
  const int Np64 = (int)round_up(N, 64);
  if (out_pair) {
    ensure(ws_x3h_, (size_t)Mp * 2 * Np64 * 2);
    PF_CHECK((void*)ws_x3h_.p != (void*)a2, PF_ERR_UNSUPPORTED, "gemm32: chained operand pairs");
  }
  if (x3_one_ && pair_ok) {
    // ONE launch: the K loop walks the cross terms first ([hi_x | lo'_x] x [lo'_W | hi_W], depth 2 Kp), scales the accumulators
    // by 2^-11, steps the cursors back (A to hi_x, W to hi_W) and adds hi_x hi_W^T (depth Kp) on top — small terms first, one
    // fp32 accumulator, no [M, N] intermediate written and read back (FFN-up: 2 x 131 MB per layer), half the launches
    GemmArgs c{};
    c.A = a2; c.lda = 2 * Kp; c.W = wcat; c.ldw = 2 * Kp; c.bias = bias; c.M = M; c.N = N; c.K = 3 * Kp;
    c.k_wrap = 2 * Kp / 64; c.a_wrap = 2 * Kp; c.w_wrap = Kp; c.wrap_scale = 1.0f / 2048.0f;
    if (out_pair) {
      c.out_f16 = (half_t*)ws_x3h_.p; c.ldc16 = 2 * Np64; c.f16_lo_off = Np64;
      x3_pair_live_ = true; x3_pair_M_ = M; x3_pair_K_ = N;
    } else {
      c.out_f32 = out; c.ldc32 = ldc;
    }
    c.scale_cols = part_scale ? scale_cols : (scale_cols ? (int)round_up(N, 64) : 0); c.scale = scale;
    c.add2 = resid2; c.ld2 = ldr; c.resid = resid; c.ldr = ldr; c.relu = relu ? 1 : 0;
    c.out_padded = 1; c.small_ws = small_ws_;
    launch_gemm(stream_, c);                                          // out = (x W^T + bias) [* scale] [+ resid2] [+ resid]; ReLU
    return;
  }
  GemmArgs g{};
  g.A = a2; g.lda = 2 * Kp; g.W = wcat + Kp; g.ldw = 2 * Kp; g.bias = bias; g.M = M; g.N = N; g.K = Kp;
  g.out_f32 = t; g.ldc32 = ldt; g.scale_cols = scale_cols ? (int)round_up(N, 64) : 0; g.scale = scale;
  g.add2 = resid2; g.ld2 = ldr;
  g.out_padded = 1; g.small_ws = small_ws_;
  launch_gemm(stream_, g);                                            // t = (hi_x hi_W^T + bias) [* scale] [+ resid2]
  GemmArgs c{};
  c.A = a2; c.lda = 2 * Kp; c.W = wcat; c.ldw = 2 * Kp; c.M = M; c.N = N; c.K = 2 * Kp;
  if (out_pair) {
    c.out_f16 = (half_t*)ws_x3h_.p; c.ldc16 = 2 * Np64; c.f16_lo_off = Np64;
    x3_pair_live_ = true; x3_pair_M_ = M; x3_pair_K_ = N;
  } else {
    c.out_f32 = out; c.ldc32 = ldc;
  }
  c.add2 = t; c.ld2 = ldt; c.resid = resid; c.ldr = ldr; c.relu = relu ? 1 : 0;
  c.scale_cols = (int)round_up(N, 64); c.scale = (scale_cols ? scale : 1.f) * (1.0f / 2048.0f);
  c.out_padded = 1; c.small_ws = small_ws_;
  launch_gemm(stream_, c);                                            // out = cross * 2^-11 [* scale] + t + resid; ReLU
}

void Engine::enc_layer_fp32(const EncLayer& L, bool first, const float* speech_dev, int B, int T, float** f) {
  const int D = mc_.d_model, M = B * T, F = mc_.ffn, Fd = mc_.feat_dim;
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  const int din = first ? Fd : D;
  if (first) {
    launch_posenc_f32(stream_, speech_dev, (const float*)ws_pe_.p, B, T, Fd, std::sqrt((float)D), f[F_T]);
    launch_layernorm(stream_, f[F_T], M, Fd, L.norm1.g, L.norm1.b, nullptr, 0, f[F_XN], Fd);
  } else {
    layernorm32(f[F_X], M, D, L.norm1, f[F_XN]);
  }
  const float* Wq = L.qkv.w32;
  // math_mode 3 above the short-input threshold: Q | K | V as ONE x3 product of N = 3 D (the weight is stored [Q | K | V] rows;
  // q-scale on the first D columns) into the three consecutive buffers read as one [M, 3 D] matrix
  const bool qkv_one = x3_mode_ && x3_one_ && x3_fuse_ && M > gemm_small_max_rows() && mc_.kernel == 11 && D % 64 == 0 &&
                       f[F_K] == f[F_Q] + (size_t)M * D && f[F_V] == f[F_K] + (size_t)M * D;
  cls32_ = "gemm32_qkv";
  if (qkv_one) {
    float* qkv = f[F_Q];
    gemm32(f[F_XN], din, Wq, din, L.qkv.bias, M, 3 * D, din, qkv, 3 * D, nullptr, 0, false, D, qscale);
    prof_begin("fsmn", 0);
    launch_fsmn_f32_ld(stream_, qkv + 2 * D, 3 * D, L.fsmn_wT, B, T, D, mc_.kernel, f[F_FS]);
    prof_end("fsmn");
    attention32(qkv, (int64_t)T * 3 * D, 3 * D, qkv + D, (int64_t)T * 3 * D, 3 * D, qkv + 2 * D, (int64_t)T * 3 * D, 3 * D, f[F_CTX],
                (int64_t)T * D, D, B, mc_.heads, T, T, true);
  } else {
  gemm32(f[F_XN], din, Wq, din, L.qkv.bias, M, D, din, f[F_Q], D, nullptr, 0, false, D, qscale);
  gemm32(f[F_XN], din, Wq + (size_t)D * din, din, L.qkv.bias + D, M, D, din, f[F_K], D, nullptr, 0, false, 0, 1.f, kX3SameInput);
  gemm32(f[F_XN], din, Wq + (size_t)2 * D * din, din, L.qkv.bias + 2 * D, M, D, din, f[F_V], D, nullptr, 0, false, 0, 1.f, kX3SameInput);
  launch_fsmn_f32(stream_, f[F_V], L.fsmn_wT, nullptr, B, T, D, mc_.kernel, f[F_FS]);
  attention32(f[F_Q], (int64_t)T * D, D, f[F_K], (int64_t)T * D, D, f[F_V], (int64_t)T * D, D, f[F_CTX],
                       (int64_t)T * D, D, B, mc_.heads, T, T, true);
  }
  cls32_ = "gemm32_out";
  if (first) {
    gemm32(f[F_CTX], D, L.out.w32, D, L.out.bias, M, D, D, f[F_X], D, f[F_FS], D, false, 0, 1.f);
  } else {
    // x = x + (lin + fsmn): the fp32 graph adds the attention block's two terms first; here the sum of three is formed in the
    // product's epilogue as (lin + fsmn) + x up to one rounding of the association
    gemm32(f[F_CTX], D, L.out.w32, D, L.out.bias, M, D, D, f[F_X], D, f[F_X], D, false, 0, 1.f, 0, f[F_FS]);
  }
  layernorm32(f[F_X], M, D, L.norm2, f[F_XN]);
  cls32_ = "gemm32_ffn1";
  gemm32(f[F_XN], D, L.w1.w32, D, L.w1.bias, M, F, D, f[F_H], F, nullptr, 0, true, 0, 1.f, kX3OutPair);
  cls32_ = "gemm32_ffn2";
  gemm32(f[F_H], F, L.w2.w32, F, L.w2.bias, M, D, F, f[F_X], D, f[F_X], D, false, 0, 1.f, kX3InPair);
  cls32_ = "gemm32_misc";
}

void Engine::forward_fp32(const float* speech_dev, int B, int T, bool want_logits) {
  const int D = mc_.d_model, F = mc_.ffn, V = mc_.vocab, Fd = mc_.feat_dim, M = B * T, T1 = T + 1;
  const int taps = mc_.cif_l_order + mc_.cif_r_order + 1;
  build_pe(T);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t W = (size_t)std::max(std::max(Fd, D), taps * D);
  size_t o_f[F_COUNT];
  o_f[F_X] = carve((size_t)M * D * 4); o_f[F_XN] = carve((size_t)M * W * 4); o_f[F_Q] = carve((size_t)M * D * 4);
  o_f[F_K] = carve((size_t)M * D * 4); o_f[F_V] = carve((size_t)M * D * 4); o_f[F_CTX] = carve((size_t)M * D * 4);
  o_f[F_FS] = carve((size_t)M * D * 4); o_f[F_H] = carve((size_t)M * F * 4); o_f[F_T] = carve((size_t)M * W * 4);
  const size_t o_H = carve((size_t)M * D * 4), o_al = carve((size_t)B * T1 * 4), o_fc = carve((size_t)B * 4), o_tn = carve((size_t)B * 4);
  const size_t o_ff = carve((size_t)B * T1 * 4), o_wc = carve((size_t)B * T1 * 4), o_wr = carve((size_t)B * T1 * 4), o_mx = carve(256);
  ensure(ws_f32_, off);
  char* base = (char*)ws_f32_.p;
  float* f[F_COUNT];
  for (int i = 0; i < F_COUNT; ++i) f[i] = (float*)(base + o_f[i]);
  H32_ = (float*)(base + o_H); alphas_ = (float*)(base + o_al);
  plan_.fire_count = (int32_t*)(base + o_fc); plan_.token_num = (int32_t*)(base + o_tn);
  plan_.fire_frame = (int32_t*)(base + o_ff); plan_.w_cur = (float*)(base + o_wc);
  plan_.w_rem = (float*)(base + o_wr); plan_.max_count = (int32_t*)(base + o_mx);

  x3_pair_live_ = false; x3a_src_ = nullptr; x3a_pair_only_ = false;      // (a forward that threw may have left an operand pair announced)
  for (size_t i = 0; i < enc_.size(); ++i) enc_layer_fp32(enc_[i], i == 0, speech_dev, B, T, f);
  if (tp_.empty()) {
    launch_layernorm(stream_, f[F_X], M, D, enc_after_.g, enc_after_.b, nullptr, 0, H32_, D);
  } else {
    launch_layernorm(stream_, f[F_X], M, D, enc_after_.g, enc_after_.b, nullptr, 0, f[F_X], D);
    for (size_t i = 0; i < tp_.size(); ++i) enc_layer_fp32(tp_[i], false, nullptr, B, T, f);
    launch_layernorm(stream_, f[F_X], M, D, tp_norm_.g, tp_norm_.b, nullptr, 0, H32_, D);
  }
  const int ldV = (int)round_up(V, 4);
  last_.peak_len = 0;
  last_.cif_peak.clear();
  if (mc_.kind == "sensevoicesmall") {
    size_t o2 = 0;
    auto c2 = [&](size_t bytes) { size_t o = o2; o2 += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
    const size_t o_lg = c2((size_t)M * ldV * 4), o_ids = c2((size_t)M * 8);
    ensure(ws_dec_, o2);
    logits_ = (float*)((char*)ws_dec_.p + o_lg); ids_dev_ = (int64_t*)((char*)ws_dec_.p + o_ids); logits_ld_ = ldV;
    gemm32(H32_, D, ctc_.w32, D, ctc_.bias, M, V, D, logits_, ldV, nullptr, 0, false, 0, 1.f);
    launch_argmax(stream_, logits_, M, V, ldV, want_logits ? 2 : 1, ids_dev_);
    last_.B = B; last_.L = T; last_.V = V; last_.T = T;
    last_.ids.assign((size_t)M, 0);
    last_.token_num.assign(B, T);
    last_.fire_count.assign(B, T);
    PF_HIP(hipMemcpyAsync(last_.ids.data(), ids_dev_, (size_t)M * 8, hipMemcpyDeviceToHost, stream_));
    last_flops_ = 0;
    return;
  }
  // ---- CIF predictor
  launch_im2col_f32(stream_, H32_, B, T, D, mc_.cif_l_order, mc_.cif_r_order, f[F_T]);
  cls32_ = "gemm32_cif";
  gemm32(f[F_T], taps * D, cif_conv_w32_, taps * D, cif_conv_.bias, M, D, taps * D, f[F_FS], D, nullptr, 0, true, 0, 1.f);
  launch_cif_alpha(stream_, f[F_FS], B, T, D, cif_out_w_, cif_out_b_, mc_.cif_smooth, mc_.cif_noise, mc_.cif_tail, alphas_);
  if (mc_.cif_cumsum) launch_cif_scan_cumsum(stream_, alphas_, B, T1, plan_);
  else launch_cif_scan(stream_, alphas_, B, T1, mc_.cif_threshold, plan_);
  export_plan(B);
  if (mc_.timestamp_head) timestamp_head_fp32(B, T);
  int32_t L = read_back_plan(B);
  if (mc_.timestamp_head) PF_HIP(hipStreamSynchronize(stream_));       // (the in-line head's results are read below as before)
  if (l_hook_) L = l_hook_(L);
  last_.B = B; last_.L = L; last_.V = V; last_.T = T;
  last_.ids.assign((size_t)B * L, 0);
  last_flops_ = 0;
  if (L == 0) return;
  // ---- decoder
  const int Md = B * L;
  size_t o2 = 0;
  auto c2 = [&](size_t bytes) { size_t o = o2; o2 += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_x = c2((size_t)Md * D * 4), o_xn = c2((size_t)Md * D * 4), o_h = c2((size_t)Md * F * 4), o_hn = c2((size_t)Md * F * 4);
  const size_t o_t = c2((size_t)Md * D * 4), o_tn2 = c2((size_t)Md * D * 4), o_q = c2((size_t)Md * D * 4), o_ctx = c2((size_t)Md * D * 4);
  const size_t o_kv = c2((size_t)M * 2 * D * 4), o_lg = c2((size_t)Md * ldV * 4), o_ids = c2((size_t)Md * 8);
  ensure(ws_dec_, o2);
  char* b2 = (char*)ws_dec_.p;
  float* xd = (float*)(b2 + o_x); float* xn = (float*)(b2 + o_xn); float* hd = (float*)(b2 + o_h); float* hn = (float*)(b2 + o_hn);
  float* t32 = (float*)(b2 + o_t); float* tn32 = (float*)(b2 + o_tn2); float* qd = (float*)(b2 + o_q); float* cx = (float*)(b2 + o_ctx);
  float* kv = (float*)(b2 + o_kv);
  logits_ = (float*)(b2 + o_lg); ids_dev_ = (int64_t*)(b2 + o_ids); logits_ld_ = ldV;
  if (mc_.cif_cumsum) launch_cif_gather_cumsum(stream_, H32_, alphas_, B, T, D, T1, plan_, L, xd);
  else launch_cif_gather(stream_, H32_, B, T, D, T1, plan_, L, xd);
  const bool bias_branch = mc_.seaco && n_hotwords_ > 0;
  float* e0 = nullptr;                                 // SeACo: the bias decoder also starts from the CIF embeds
  if (bias_branch) {
    ensure(ws_seaco_in_, (size_t)Md * D * 4);
    e0 = (float*)ws_seaco_in_.p;
    PF_HIP(hipMemcpyAsync(e0, xd, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  }
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  cls32_ = "gemm32_dec";
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2) {
    layernorm32(xd, Md, D, n1, xn);
    gemm32(xn, D, w1.w32, D, w1.bias, Md, F, D, hd, F, nullptr, 0, true, 0, 1.f);
    launch_layernorm(stream_, hd, Md, F, fn.g, fn.b, nullptr, 0, hn, F);
    gemm32(hn, F, w2.w32, F, nullptr, Md, D, F, t32, D, nullptr, 0, false, 0, 1.f);
  };
  for (size_t i = 0; i < dec_.size(); ++i) {
    const DecLayer& Lr = dec_[i];
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2);
    launch_layernorm(stream_, t32, Md, D, Lr.norm2.g, Lr.norm2.b, nullptr, 0, tn32, D);
    launch_fsmn_dec(stream_, tn32, Lr.fsmn_wT, plan_.token_num, B, L, D, mc_.kernel, xd);
    layernorm32(xd, Md, D, Lr.norm3, xn);
    gemm32(xn, D, Lr.q.w32, D, Lr.q.bias, Md, D, D, qd, D, nullptr, 0, false, D, qscale);
    gemm32(H32_, D, Lr.kv32.w32, D, Lr.kv32.bias, M, 2 * D, D, kv, 2 * D, nullptr, 0, false, 0, 1.f);
    attention32(qd, (int64_t)L * D, D, kv, (int64_t)T * 2 * D, 2 * D, kv + D, (int64_t)T * 2 * D, 2 * D, cx,
                         (int64_t)L * D, D, B, mc_.heads, L, T, true);
    gemm32(cx, D, Lr.out.w32, D, Lr.out.bias, Md, D, D, xd, D, xd, D, false, 0, 1.f);
  }
  ffn_dec(dec_final_norm1_, dec_final_w1_, dec_final_ffn_norm_, dec_final_w2_);
  launch_layernorm(stream_, t32, Md, D, dec_after_.g, dec_after_.b, nullptr, 0, xn, D);
  cls32_ = "gemm32_vocab";
  gemm32(xn, D, dec_out_.w32, D, dec_out_.bias, Md, V, D, logits_, ldV, nullptr, 0, false, 0, 1.f);
  launch_argmax(stream_, logits_, Md, V, ldV, want_logits ? 2 : 1, ids_dev_);
  cls32_ = "gemm32_misc";
  if (bias_branch) seaco_head_fp32(B, L, e0, xn, want_logits);      // xn = the ASR decoder's after_norm hidden
  PF_HIP(hipMemcpyAsync(last_.ids.data(), ids_dev_, (size_t)Md * 8, hipMemcpyDeviceToHost, stream_));
}

// ---- fp32 forms of the two heads (math_mode 1): the same graphs as timestamp_head / seaco_head with fp32 weights and
// activations, one launch per graph node; the recurrences run as one fp32 GEMM + one cell kernel per time step.
void Engine::lstm_fp32(const float* x, int Bn, int Tn, const float* w_ih, const float* w_hh, const float* bias, bool reverse,
                       float* xg, float* gates, float* hbuf, float* cbuf, float* hout, int ldh, int col0) {
  const int D = mc_.d_model;
  // input half of the gates for every row at once: xg[b * Tn + t, 0:4D] = x W_ih^T + (b_ih + b_hh)
  gemm32(x, D, w_ih, D, bias, Bn * Tn, 4 * D, D, xg, 4 * D, nullptr, 0, false, 0, 1.f);
  PF_HIP(hipMemsetAsync(hbuf, 0, (size_t)Bn * D * 4, stream_));
  PF_HIP(hipMemsetAsync(cbuf, 0, (size_t)Bn * D * 4, stream_));
  for (int st = 0; st < Tn; ++st) {
    const int t = reverse ? Tn - 1 - st : st;
    // gates[b] = xg[b * Tn + t] + h[b] W_hh^T   (rows of the residual operand are Tn * 4D apart)
    launch_gemm_f32(stream_, hbuf, D, w_hh, D, nullptr, Bn, 4 * D, D, gates, 4 * D, xg + (size_t)t * 4 * D, Tn * 4 * D, false, 0, 1.f);
    launch_lstm_cell_f32(stream_, gates, 4 * D, cbuf, hbuf, hout + (size_t)t * ldh + col0, (int64_t)Tn * ldh, Bn, D);
  }
}

void Engine::timestamp_head_fp32(int B, int T) {
  const int D = mc_.d_model, up = mc_.upsample;
  const int M = B * T, T3 = up * T;
  const int64_t M3 = (int64_t)B * T3;
  PF_CHECK(ts_up_w32_, PF_ERR_UNSUPPORTED, "fp32 timestamp head: operands were not prepared (engine not created in math_mode 1)");
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_up = carve((size_t)M3 * D * 4), o_xg = carve((size_t)M3 * 4 * D * 4), o_ho = carve((size_t)M3 * 2 * D * 4);
  const size_t o_g = carve((size_t)B * 4 * D * 4), o_h = carve((size_t)B * D * 4), o_c = carve((size_t)B * D * 4);
  const size_t o_al = carve((size_t)M3 * 4), o_pk = carve((size_t)M3 * 4);
  // math_mode 3: the recurrence as ONE persistent launch with (hi, lo') pair operands (k_bicif.hip, lstm_ring_kernel<true>): room for
  // both directions' input gates, the four pair slots of h and the sync words
  const bool x3_rec = x3_mode_ && x3_fuse_ && !lstm_steps_ && D == 512 && (D / 8) * 2 * ((B + 31) / 32) <= cus_;
  const size_t o_xg2 = x3_rec ? carve((size_t)(M3 + 256) * 8 * D * 4) : 0, o_hs = x3_rec ? carve((size_t)2 * 4 * B * 2 * D * 2) : 0, o_sw = x3_rec ? carve(256) : 0;
  ensure(ws_ts_, off);
  char* base = (char*)ws_ts_.p;
  float* up32 = (float*)(base + o_up); float* xg = (float*)(base + o_xg); float* hout = (float*)(base + o_ho);
  float* gates = (float*)(base + o_g); float* hb = (float*)(base + o_h); float* cb = (float*)(base + o_c);
  float* al = (float*)(base + o_al);
  us_peak_ = (float*)(base + o_pk);
  // [M, 3D] row-major IS [3M, D]
  gemm32(H32_, D, ts_up_w32_, D, ts_up_.bias, M, up * D, D, up32, up * D, nullptr, 0, false, 0, 1.f);
  const char* sfx[2] = {"", "_reverse"};
  bool done = false;
  if (x3_rec) {
    if (!ts_whh_x3_) {                                   // [2 dir][4D][hi (D) | lo' (D)], built once from the fp32 tensors
      ts_whh_x3_ = (half_t*)dalloc((size_t)2 * 4 * D * 2 * D * 2);
      for (int d = 0; d < 2; ++d)
        launch_split_x3(stream_, tensor(std::string("predictor.blstm.weight_hh") + sfx[d]).dev, 4 * D, D, D, ts_whh_x3_ + (size_t)d * 4 * D * 2 * D, 2 * D, D, 0);
    }
    float* xg2 = (float*)(base + o_xg2);
    for (int d = 0; d < 2; ++d)                          // input half of the gates, both directions side by side: [M3, 8D]
      gemm32(up32, D, tensor(std::string("predictor.blstm.weight_ih") + sfx[d]).dev, D, ts_ih_.bias + (size_t)d * 4 * D, (int)M3, 4 * D, D,
             xg2 + (size_t)d * 4 * D, 8 * D, nullptr, 0, false, 0, 1.f, d == 1 ? kX3SameInput : 0);
    LstmArgs a{};
    a.whh = ts_whh_x3_; a.xg = xg2; a.hstate = (half_t*)(base + o_hs); a.cstate = nullptr; a.hout = hout; a.B = B; a.T3 = T3; a.D = D; a.ndir = 2;
    unsigned* sw = (unsigned*)(base + o_sw);
    done = launch_lstm_persistent_x3(stream_, a, sw);
    if (done) lstm_err_ = sw + 63;
  }
  if (!done)
  for (int d = 0; d < 2; ++d)
    lstm_fp32(up32, B, T3, tensor(std::string("predictor.blstm.weight_ih") + sfx[d]).dev,
              tensor(std::string("predictor.blstm.weight_hh") + sfx[d]).dev, ts_ih_.bias + (size_t)d * 4 * D, d == 1, xg, gates, hb,
              cb, hout, 2 * D, d * D);
  launch_us_alpha(stream_, hout, M3, 2 * D, ts_out_w_, ts_out_b_, mc_.cif_smooth2, mc_.cif_noise2, al);
  launch_us_peak(stream_, al, plan_.token_num, B, T3, mc_.cif_threshold - 1e-4f, us_peak_);
  last_.peak_len = T3;
  last_.cif_peak.resize((size_t)M3);
  PF_HIP(hipMemcpyAsync(last_.cif_peak.data(), us_peak_, (size_t)M3 * 4, hipMemcpyDeviceToHost, stream_));
}

void Engine::seaco_head_fp32(int B, int L, const float* e0, const float* hid_asr, bool want_logits) {
  const int D = mc_.d_model, V = mc_.vocab, Fs = mc_.seaco_ffn, ns = (int)sdec_.size();
  const int N = n_hotwords_, J = 10, NJ = N * J;
  const int Md = B * L, R = 2 * Md;
  const int ldV = (int)round_up(V, 4);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlign); return o; };
  const size_t o_ids = carve((size_t)NJ * 4), o_e = carve((size_t)NJ * D * 4), o_e2 = carve((size_t)NJ * D * 4), o_xg = carve((size_t)NJ * 4 * D * 4);
  const size_t o_g = carve((size_t)N * 4 * D * 4), o_hb = carve((size_t)N * D * 4), o_cb = carve((size_t)N * D * 4);
  const size_t o_kv = carve((size_t)NJ * 2 * D * 4), o_x = carve((size_t)R * D * 4), o_xn = carve((size_t)R * D * 4);
  const size_t o_h = carve((size_t)R * Fs * 4), o_hn = carve((size_t)R * Fs * 4), o_t = carve((size_t)R * D * 4), o_tn = carve((size_t)R * D * 4);
  const size_t o_q = carve((size_t)R * D * 4), o_cx = carve((size_t)R * D * 4), o_hid = carve((size_t)R * D * 4);
  const size_t o_dha = carve((size_t)Md * ldV * 4), o_did = carve((size_t)Md * 8), o_tn2 = carve((size_t)2 * B * 4);
  ensure(ws_seaco_, off);
  char* base = (char*)ws_seaco_.p;
  int32_t* ids = (int32_t*)(base + o_ids);
  float* ea = (float*)(base + o_e); float* eb = (float*)(base + o_e2); float* xg = (float*)(base + o_xg);
  float* gates = (float*)(base + o_g); float* hb = (float*)(base + o_hb); float* cb = (float*)(base + o_cb);
  float* kv = (float*)(base + o_kv); float* xs = (float*)(base + o_x); float* xn = (float*)(base + o_xn);
  float* hd = (float*)(base + o_h); float* hn = (float*)(base + o_hn); float* t32 = (float*)(base + o_t); float* tn32 = (float*)(base + o_tn);
  float* qd = (float*)(base + o_q); float* cx = (float*)(base + o_cx); float* hid = (float*)(base + o_hid);
  float* dha = (float*)(base + o_dha); int64_t* dha_ids = (int64_t*)(base + o_did); int32_t* tn2 = (int32_t*)(base + o_tn2);
  // ---- hotword embedder: Embedding -> LSTM stack (all J outputs kept), rows n * J + j
  PF_HIP(hipMemcpyAsync(ids, hotwords_.data(), (size_t)NJ * 4, hipMemcpyHostToDevice, stream_));
  launch_embed_gather(stream_, seaco_embed_w_, ids, NJ, D, (int)tensor("seaco.embed.weight").shape[0], ea, nullptr);
  float* cur = ea;
  float* nxt = eb;
  for (size_t l = 0; l < seaco_lstm_.size(); ++l) {
    const std::string p = "seaco.lstm.l" + std::to_string(l);
    lstm_fp32(cur, N, J, tensor(p + ".weight_ih").dev, tensor(p + ".weight_hh").dev, seaco_lstm_[l].ih.bias, false, xg, gates, hb, cb,
              nxt, D, 0);
    std::swap(cur, nxt);
  }
  const float* bias_embed = cur;                       // [NJ, D]
  // ---- bias decoder on [CIF embeds ; decoder hidden]
  PF_HIP(hipMemcpyAsync(xs, e0, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  PF_HIP(hipMemcpyAsync(xs + (size_t)Md * D, hid_asr, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  PF_HIP(hipMemcpyAsync(tn2, plan_.token_num, (size_t)B * 4, hipMemcpyDeviceToDevice, stream_));
  PF_HIP(hipMemcpyAsync(tn2 + B, plan_.token_num, (size_t)B * 4, hipMemcpyDeviceToDevice, stream_));
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2) {
    launch_layernorm(stream_, xs, R, D, n1.g, n1.b, nullptr, 0, xn, D);
    gemm32(xn, D, w1.w32, D, w1.bias, R, Fs, D, hd, Fs, nullptr, 0, true, 0, 1.f);
    launch_layernorm(stream_, hd, R, Fs, fn.g, fn.b, nullptr, 0, hn, Fs);
    gemm32(hn, Fs, w2.w32, Fs, nullptr, R, D, Fs, t32, D, nullptr, 0, false, 0, 1.f);
  };
  for (int i = 0; i < ns; ++i) {
    const DecLayer& Lr = sdec_[i];
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2);
    launch_layernorm(stream_, t32, R, D, Lr.norm2.g, Lr.norm2.b, nullptr, 0, tn32, D);
    launch_fsmn_dec(stream_, tn32, Lr.fsmn_wT, tn2, 2 * B, L, D, mc_.seaco_kernel, xs);
    launch_layernorm(stream_, xs, R, D, Lr.norm3.g, Lr.norm3.b, nullptr, 0, xn, D);
    gemm32(xn, D, Lr.q.w32, D, Lr.q.bias, R, D, D, qd, D, nullptr, 0, false, D, qscale);
    gemm32(bias_embed, D, Lr.kv32.w32, D, Lr.kv32.bias, NJ, 2 * D, D, kv, 2 * D, nullptr, 0, false, 0, 1.f);
    attention32(qd, (int64_t)L * D, D, kv, 0, 2 * D, kv + D, 0, 2 * D, cx, (int64_t)L * D, D, 2 * B, mc_.heads, L, NJ);
    gemm32(cx, D, Lr.out.w32, D, Lr.out.bias, R, D, D, xs, D, xs, D, false, 0, 1.f);
  }
  ffn_dec(seaco_final_norm1_, seaco_final_w1_, seaco_final_ffn_norm_, seaco_final_w2_);
  launch_layernorm(stream_, t32, R, D, seaco_after_.g, seaco_after_.b, nullptr, 0, hid, D);
  // ---- merged = cif_attended + dec_attended -> hotword_output_layer -> NO-BIAS merge with the ASR rows
  launch_add_f32(stream_, hid, hid + (size_t)Md * D, (int64_t)Md * D);
  gemm32(hid, D, seaco_out_.w32, D, seaco_out_.bias, Md, V, D, dha, ldV, nullptr, 0, false, 0, 1.f);
  launch_argmax(stream_, dha, Md, V, ldV, 2, dha_ids);
  launch_seaco_merge(stream_, dha, ldV, dha_ids, Md, V, mc_.seaco_nobias, want_logits ? 1 : 0, logits_, logits_ld_, ids_dev_);
}

void Engine::forward_feats_host(const float* speech, int B, int T, bool want_logits) {
  PF_CHECK(speech && B > 0 && T > 0, PF_ERR_INVALID_ARG, "forward_feats: bad arguments");
  PF_HIP(hipSetDevice(device_));
  const size_t n = (size_t)B * T * mc_.feat_dim;
  ensure(ws_speech_, n * 4);
  PF_HIP(hipMemcpyAsync(ws_speech_.p, speech, n * 4, hipMemcpyHostToDevice, stream_));
  forward_device((const float*)ws_speech_.p, B, T, want_logits);
}

void Engine::model_proj_host(const float* const* speech, const int32_t* n_floats, int B, bool want_logits) {
  PF_CHECK(B > 0, PF_ERR_INVALID_ARG, "model_proj: empty batch");
  PF_HIP(hipSetDevice(device_));
  const int W = mc_.feat_dim;
  int maxf = 0;
  std::vector<int64_t> offs(B);
  int64_t tot = 0;
  for (int b = 0; b < B; ++b) {
    PF_CHECK(speech[b] || n_floats[b] == 0, PF_ERR_INVALID_ARG, "model_proj: null speech");
    offs[b] = tot;
    tot += round_up(n_floats[b], 4);
    maxf = std::max(maxf, n_floats[b]);
  }
  PF_CHECK(maxf > 0 && maxf % W == 0, PF_ERR_INVALID_ARG, "model_proj: feature length not a multiple of 560");
  const int T = maxf / W;
  ensure(ws_tmp_, (size_t)tot * 4 + (size_t)B * 12 + 64);
  float* rag = (float*)ws_tmp_.p;
  int64_t* offd = (int64_t*)((char*)ws_tmp_.p + round_up(tot * 4, 8));
  int32_t* nd = (int32_t*)(offd + B);
  for (int b = 0; b < B; ++b)
    if (n_floats[b] > 0)
      PF_HIP(hipMemcpyAsync(rag + offs[b], speech[b], (size_t)n_floats[b] * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(offd, offs.data(), (size_t)B * 8, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(nd, n_floats, (size_t)B * 4, hipMemcpyHostToDevice, stream_));
  ensure(ws_speech_, (size_t)B * maxf * 4);
  launch_pad_sentinel(stream_, rag, offd, nd, B, maxf, (float*)ws_speech_.p);
  forward_device((const float*)ws_speech_.p, B, T, want_logits);
}
}  // namespace pf
