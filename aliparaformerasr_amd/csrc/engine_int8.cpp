// engine_int8.cpp — pf_engine_config.math_mode = 2: the graph as the reference's DEFAULT model files compute it.
//
// `-accuracy int8` is the reference CLI's default (AliParaformerAsr.Examples/Program.cs:98-101 -> model.int8.onnx,
// Examples/OfflineAliParaformerAsrRecognizer.cs:17-22): FunASR exports passed through onnxruntime's quantize_dynamic,
// in which every MatMul with a constant weight runs as DynamicQuantizeLinear (per-tensor uint8 activations, computed at
// run time) + MatMulInteger (uint8 weights, per output channel) + a float rescale (k_quant.hip restates the operator
// arithmetic; oracle/int8.py is its numpy twin).  Here those products run on v_mfma_i32_32x32x32_i8 with exact int32
// accumulation (k_gemm.hip gemm_i8_pp3); everything else — LayerNorm, FSMN, attention (f16 MFMA), CIF, soft-max, arg-max
// — is the f16 path's kernels.  What is NOT quantised, as in the export: the depthwise FSMN convolutions, the CIF
// Conv1d (Conv nodes) and the N = 1 predictor output (kept fp32 here).
// Weights: a container converted from an int8 export carries the stored bytes (`<linear>.weight_q` u8, `.weight_zp` u8,
// `.weight_scale` f32, aliparaformerasr_amd/convert.py) and those are what is multiplied; a container with fp32 tensors
// only (the synthetic models) is quantised at first use on the device with quantize_dynamic's per-channel algorithm.
#include <cmath>
#include <cstring>

#include "engine.h"

namespace pf {

static const size_t kAlignQ = 256;

void Engine::ensure_q(int64_t rows, int kpad) {
  const int64_t rp = round_up(rows, 256) + 256;
  const size_t need = (size_t)rp * kpad + (size_t)rp * 4 + 1024;
  if (!ws_q_.p || ws_q_.bytes < need || q_kpad_ < kpad || q_rows_ < rp) {
    const int64_t r2 = std::max(rp, q_rows_);
    const int k2 = std::max(kpad, q_kpad_);
    ensure(ws_q_, (size_t)r2 * k2 + (size_t)r2 * 4 + 1024);
    q_rows_ = r2; q_kpad_ = k2;
    char* b = (char*)ws_q_.p;
    q_a_ = (int8_t*)b;
    q_rowsum_ = (int32_t*)(b + round_up((int64_t)r2 * k2, (int64_t)kAlignQ));
    q_params_ = (float*)((char*)q_rowsum_ + (size_t)r2 * 4);
    q_scratch_ = (unsigned*)(q_params_ + 8);
  }
}

const QLin& Engine::qlin_raw(const float* w32, const float* bias, int N, int K) {
  auto it = qlins_.find(w32);
  if (it != qlins_.end()) return it->second;
  QLin q;
  q.N = N; q.K = K; q.Kpad = (int)round_up(K, 128);
  const size_t npad = (size_t)round_up(N, 128);
  q.w = (int8_t*)dalloc(npad * q.Kpad);
  PF_HIP(hipMemsetAsync(q.w, 0, npad * q.Kpad, stream_));
  q.colsum = (int32_t*)dalloc((size_t)(N + 8) * 4);
  q.wzp = (int32_t*)dalloc((size_t)(N + 8) * 4);
  q.wscale = (float*)dalloc((size_t)(N + 8) * 4);
  PF_HIP(hipMemsetAsync(q.colsum, 0, (size_t)(N + 8) * 4, stream_));
  PF_HIP(hipMemsetAsync(q.wzp, 0, (size_t)(N + 8) * 4, stream_));
  PF_HIP(hipMemsetAsync(q.wscale, 0, (size_t)(N + 8) * 4, stream_));
  // an int8 export's container carries the stored bytes beside the float image: multiply THOSE (no re-quantisation)
  const auto nm = lin_names_.find(w32);
  const Tensor* sq = nm == lin_names_.end() ? nullptr : tensor_u8(nm->second + ".weight_q");
  if (sq) {
    const Tensor* sz = tensor_u8(nm->second + ".weight_zp");
    const Tensor& ss = tensor(nm->second + ".weight_scale");
    PF_CHECK(sz && sq->shape.size() == 2 && sq->shape[0] == N && sq->shape[1] == K && sz->numel == N && ss.numel == N, PF_ERR_FORMAT,
             "weights: '" + nm->second + "' weight_q / weight_zp / weight_scale do not match the Linear's [N, K]");
    launch_import_weight(stream_, (const uint8_t*)sq->dev, (const uint8_t*)sz->dev, ss.dev, N, K, q.w, q.Kpad, q.colsum, q.wzp, q.wscale);
  } else {
    launch_quantize_weight(stream_, w32, N, K, q.w, q.Kpad, q.colsum, q.wzp, q.wscale);
  }
  q.dz = (int32_t*)dalloc((size_t)(N + 8) * 4);
  PF_HIP(hipMemsetAsync(q.dz, 0, (size_t)(N + 8) * 4, stream_));
  launch_pack_dz(stream_, q.colsum, q.wzp, N, K, q.dz);
  q.bias = bias;
  return qlins_.emplace(w32, q).first->second;
}

const QLin& Engine::qlin(const Lin& l, bool bias) {
  PF_CHECK(l.w32, PF_ERR_UNSUPPORTED, "int8 mode: a Linear without its fp32 tensor");
  return qlin_raw(l.w32, bias ? l.bias : nullptr, l.N, l.K);
}

void Engine::quantize_act(const QAct& dst, int kpad, const float* x32, const half_t* x16, int ldx, int64_t M, int K, const LNp* ln,
                          bool have_range) {
  if (!q_part_) q_part_ = (float*)dalloc(quant_scratch_bytes());
  prof_begin("quantize", 0);
  if (have_range) {                                    // pass 1 was the producing GEMM's epilogue
    PF_CHECK(!ln, PF_ERR_INVALID_ARG, "quantize_act: a LayerNorm input has no producer range");
    launch_quantize(stream_, x32, x16, M, K, ldx, dst.a, kpad, dst.rowsum, dst.params, q_part_);
    prof_end("quantize");
    return;
  }
  if (ln) {                                            // LayerNorm(x32) is never stored: normalised in registers by both passes
    PF_CHECK(x32 && ln->D == K && ldx == K, PF_ERR_INVALID_ARG, "quantize_act: LayerNorm rows must be dense fp32");
    launch_ln_minmax(stream_, x32, M, K, ln->g, ln->b, q_part_);
    launch_ln_quantize(stream_, x32, M, K, ln->g, ln->b, dst.a, kpad, dst.rowsum, dst.params, q_part_);
  } else {
    launch_minmax(stream_, x32, x16, M, K, ldx, q_part_);
    launch_quantize(stream_, x32, x16, M, K, ldx, dst.a, kpad, dst.rowsum, dst.params, q_part_);
  }
  prof_end("quantize");
}

// Is this Linear a DynamicQuantizeLinear + MatMulInteger pair in the model file?  A container made from an int8 export
// answers by what it carries: stored bytes (`<linear>.weight_q`) = quantised, none = the export left that MatMul in
// float (FunASR passes nodes_to_exclude to quantize_dynamic: the vocabulary projections and the N = 1 predictor outputs).
// A container with fp32 tensors only (the synthetic models) quantises everything except the names listed in its
// `int8_exclude` config key.
bool Engine::lin_quantised(const Lin& l) const {
  const auto nm = lin_names_.find(l.w32);
  if (nm == lin_names_.end()) return !any_stored_q_;
  if (any_stored_q_) return tensors_.count(nm->second + ".weight_q") != 0;
  for (const std::string& e : int8_exclude_)
    if (nm->second.compare(0, e.size(), e) == 0) return false;
  return true;
}

// a Linear the export did not quantise, inside math_mode 2: the f16 path's GEMM (A rounded to f16 as everywhere in mode 0)
void Engine::fgemm_in_int8(const char* cls, const Lin& l, bool bias, const float* x32, const half_t* x16, int ldx, int M, float* out32,
                           int ld32, half_t* out16, int ld16, const float* resid, int ldr, const float* add2, int ld2, bool relu,
                           int scale_cols, float scale, const LNp* ln, int range) {
  PF_CHECK(l.w, PF_ERR_UNSUPPORTED, "int8 mode: an un-quantised Linear without its f16 operand");
  const half_t* A = x16;
  int lda = ldx;
  if (!A) {
    PF_CHECK(x32, PF_ERR_INVALID_ARG, "int8 mode: an un-quantised Linear needs its input");
    const int64_t Mp = round_up(M, 256) + 256;
    ensure(ws_qf_, (size_t)Mp * l.Kpad * 2);
    half_t* a16 = (half_t*)ws_qf_.p;
    // the conversion writes K of Kpad columns per row: the pad columns meet zero weight columns in the GEMM, but the
    // arena is re-carved per call and a stale Inf / NaN bit pattern times zero is NaN — clear them
    if (l.K != l.Kpad) PF_HIP(hipMemsetAsync(a16, 0, (size_t)Mp * l.Kpad * 2, stream_));
    prof_begin("layernorm", 0);
    if (ln) launch_layernorm(stream_, x32, M, ln->D, ln->g, ln->b, a16, l.Kpad, nullptr, 0);
    else launch_f32_to_f16(stream_, x32, M, l.K, ldx, a16, l.Kpad);
    prof_end("layernorm");
    A = a16; lda = l.Kpad;
  }
  gemm(cls, l, A, lda, M, out32, ld32, out16, ld16, resid, ldr, add2, ld2, relu, scale_cols, scale, bias);
  if (range & kRangeOut) {                             // the consumer's quantiser expects pass 1 done
    if (!q_part_) q_part_ = (float*)dalloc(quant_scratch_bytes());
    launch_minmax(stream_, out16 ? nullptr : out32, out16, M, l.N, out16 ? ld16 : ld32, q_part_);
  }
}

void Engine::qgemm(const char* cls, const Lin& lw, bool use_bias, const float* x32, const half_t* x16, int ldx, int M, float* out32, int ld32,
                   half_t* out16, int ld16, const float* resid, int ldr, const float* add2, int ld2, bool relu, int scale_cols,
                   float scale, const LNp* ln, const QAct* pre, int range) {
  if (M == 0) return;
  if (!lin_quantised(lw)) {
    PF_CHECK(!pre, PF_ERR_UNSUPPORTED, "int8 mode: a pre-quantised input for an un-quantised Linear");
    fgemm_in_int8(cls, lw, use_bias, x32, x16, ldx, M, out32, ld32, out16, ld16, resid, ldr, add2, ld2, relu, scale_cols, scale, ln, range);
    return;
  }
  const QLin& w = qlin(lw, use_bias);
  QAct act;
  if (pre) {
    act = *pre;                                        // quantised earlier (the encoder memory: one tensor, sixteen K / V projections)
  } else {
    ensure_q(M, w.Kpad);
    act.a = q_a_; act.rowsum = q_rowsum_; act.params = q_params_;
    if (x32 || x16) quantize_act(act, w.Kpad, x32, x16, ldx, M, w.K, ln, (range & kRangeIn) != 0);   // null / null: the previous call's tensor is reused
  }
  prof_begin(cls, 2.0 * M * (double)w.N * w.K);
  GemmI8Args g{};
  g.A = act.a; g.lda = w.Kpad; g.W = w.w; g.ldw = w.Kpad;
  g.rowsum = act.rowsum; g.colsum = w.colsum; g.wzp = w.wzp; g.wscale = w.wscale; g.aparams = act.params;
  g.bias = w.bias; g.M = M; g.N = w.N; g.K = w.K; g.Kpad = w.Kpad;
  g.out_f32 = out32; g.ldc32 = ld32; g.out_f16 = out16; g.ldc16 = ld16;
  g.resid = resid; g.ldr = ldr; g.add2 = add2; g.ld2 = ld2;
  g.relu = relu ? 1 : 0; g.scale_cols = scale_cols; g.scale = scale;
  g.dz = w.dz; g.out_padded = 1;                       // every f16 result of this path lands in a buffer padded to whole tiles
  if (range & kRangeOut) {
    if (!q_part_) q_part_ = (float*)dalloc(quant_scratch_bytes());
    g.range_out = q_part_;
  }
  launch_gemm_i8(stream_, g);
  prof_end(cls);
}

void Engine::forward_int8(const float* speech_dev, int B, int T, bool want_logits) {
  const int D = mc_.d_model, F = mc_.ffn, V = mc_.vocab, Fd = mc_.feat_dim, T1 = T + 1;
  const int64_t M = (int64_t)B * T, Mp = round_up(M, 128) + 128;
  const int taps = mc_.cif_l_order + mc_.cif_r_order + 1;
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  build_pe(T);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlignQ); return o; };
  const size_t o_x = carve(Mp * D * 4), o_t = carve(Mp * std::max(Fd, D) * 4);
  const size_t o_qkv = carve(Mp * 3 * D * 2), o_ctx = carve(Mp * D * 2), o_fsm = carve(Mp * D * 4);
  const size_t o_h = carve(Mp * std::max(F, taps * D) * 2), o_H32 = carve(Mp * D * 4), o_H16 = carve(Mp * D * 2);
  const size_t o_al = carve((size_t)B * T1 * 4), o_fc = carve((size_t)B * 4), o_tn = carve((size_t)B * 4);
  const size_t o_ff = carve((size_t)B * T1 * 4), o_wc = carve((size_t)B * T1 * 4), o_wr = carve((size_t)B * T1 * 4), o_mx = carve(256);
  ensure(ws_enc_, off);
  char* base = (char*)ws_enc_.p;
  x_ = (float*)(base + o_x);
  float* t32e = (float*)(base + o_t);
  qkv16_ = (half_t*)(base + o_qkv); ctx16_ = (half_t*)(base + o_ctx); fsm_ = (float*)(base + o_fsm); h16_ = (half_t*)(base + o_h);
  H32_ = (float*)(base + o_H32); H16_ = (half_t*)(base + o_H16); alphas_ = (float*)(base + o_al);
  plan_.fire_count = (int32_t*)(base + o_fc); plan_.token_num = (int32_t*)(base + o_tn);
  plan_.fire_frame = (int32_t*)(base + o_ff); plan_.w_cur = (float*)(base + o_wc);
  plan_.w_rem = (float*)(base + o_wr); plan_.max_count = (int32_t*)(base + o_mx);

  // ---- encoder: LayerNorm (fp32) -> quantise -> QKV (f16 out, q scaled) -> attention (f16) | FSMN(V) -> quantise(ctx) ->
  //      out-projection (+ FSMN memory + residual, fp32) -> LayerNorm -> quantise -> FFN-up + ReLU (f16) -> quantise -> FFN-down + residual
  auto layer = [&](const EncLayer& L, bool first) {
    const int din = first ? Fd : D;
    if (first) launch_posenc_f32(stream_, speech_dev, (const float*)ws_pe_.p, B, T, Fd, std::sqrt((float)D), t32e);
    qgemm("gemm_qkv", L.qkv, true, first ? t32e : x_, nullptr, din, (int)M, nullptr, 0, qkv16_, 3 * D, nullptr, 0, nullptr, 0, false, D, qscale,
          &L.norm1);
    AttnArgs a{};
    a.q = qkv16_; a.k = qkv16_ + D; a.v = qkv16_ + 2 * D; a.o = ctx16_;
    a.q_bstride = a.k_bstride = a.v_bstride = (int64_t)T * 3 * D;
    a.q_rstride = a.k_rstride = a.v_rstride = 3 * D;
    a.o_bstride = (int64_t)T * D; a.o_rstride = D;
    a.B = B; a.H = mc_.heads; a.Lq = T; a.Lk = T;
    if (!q_part_) q_part_ = (float*)dalloc(quant_scratch_bytes());
    const bool ctx_range = attention_reports_range(a);   // the context's {min, max} from the attention epilogue: no min / max pass
    a.range = ctx_range ? q_part_ : nullptr;
    prof_begin("attn_self", 4.0 * B * (double)T * T * D);
    launch_attention(stream_, a);
    prof_end("attn_self");
    prof_begin("fsmn", 0);
    launch_fsmn_enc(stream_, qkv16_ + 2 * D, 3 * D, L.fsmn_wT, B, T, D, mc_.kernel, fsm_);
    prof_end("fsmn");
    qgemm("gemm_out", L.out, true, nullptr, ctx16_, D, (int)M, x_, D, nullptr, 0, first ? nullptr : x_, D, fsm_, D, false, 0, 1.f, nullptr,
          nullptr, ctx_range ? kRangeIn : 0);
    qgemm("gemm_ffn1", L.w1, true, x_, nullptr, D, (int)M, nullptr, 0, h16_, F, nullptr, 0, nullptr, 0, true, 0, 1.f, &L.norm2, nullptr, kRangeOut);
    qgemm("gemm_ffn2", L.w2, true, nullptr, h16_, F, (int)M, x_, D, nullptr, 0, x_, D, nullptr, 0, false, 0, 1.f, nullptr, nullptr, kRangeIn);
  };
  for (size_t i = 0; i < enc_.size(); ++i) layer(enc_[i], i == 0);
  prof_begin("layernorm", 0);
  if (tp_.empty()) {
    launch_layernorm(stream_, x_, M, D, enc_after_.g, enc_after_.b, H16_, D, H32_, D);
  } else {
    launch_layernorm(stream_, x_, M, D, enc_after_.g, enc_after_.b, nullptr, 0, x_, D);
  }
  prof_end("layernorm");
  if (!tp_.empty()) {
    for (size_t i = 0; i < tp_.size(); ++i) layer(tp_[i], false);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, x_, M, D, tp_norm_.g, tp_norm_.b, H16_, D, H32_, D);
    prof_end("layernorm");
  }
  const int ldV = (int)round_up(V, 4);
  last_flops_ = 0;
  if (mc_.kind == "sensevoicesmall") {
    size_t o2 = 0;
    auto c2 = [&](size_t bytes) { size_t o = o2; o2 += round_up((int64_t)bytes, (int64_t)kAlignQ); return o; };
    const size_t o_lg = c2((size_t)Mp * ldV * 4), o_ids = c2((size_t)M * 8);
    ensure(ws_dec_, o2);
    logits_ = (float*)((char*)ws_dec_.p + o_lg); ids_dev_ = (int64_t*)((char*)ws_dec_.p + o_ids); logits_ld_ = ldV;
    qgemm("gemm_vocab", ctc_, true, H32_, nullptr, D, (int)M, logits_, ldV, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
    prof_begin("argmax", 0);
    launch_argmax(stream_, logits_, M, V, ldV, want_logits ? 2 : 1, ids_dev_);
    prof_end("argmax");
    last_.B = B; last_.L = T; last_.V = V; last_.T = T;
    last_.ids.assign((size_t)M, 0);
    last_.token_num.assign(B, T);
    last_.fire_count.assign(B, T);
    PF_HIP(hipMemcpyAsync(last_.ids.data(), ids_dev_, (size_t)M * 8, hipMemcpyDeviceToHost, stream_));
    return;
  }
  // ---- CIF predictor: the Conv1d is a Conv node (not quantised by quantize_dynamic): the f16 path's im2col GEMM
  if (ev_enc_) PF_HIP(hipEventRecord(ev_enc_, stream_));
  last_.peak_len = 0;
  last_.cif_peak.clear();
  {
    half_t* col16 = h16_;
    float* conv32 = fsm_;
    prof_begin("cif_misc", 0);
    launch_cif_im2col(stream_, H16_, B, T, D, mc_.cif_l_order, mc_.cif_r_order, col16);
    prof_end("cif_misc");
    gemm("gemm_cif", cif_conv_, col16, taps * D, (int)M, conv32, D, nullptr, 0, nullptr, 0, nullptr, 0, true, 0, 1.f);
    prof_begin("cif_misc", 0);
    launch_cif_alpha(stream_, conv32, B, T, D, cif_out_w_, cif_out_b_, mc_.cif_smooth, mc_.cif_noise, mc_.cif_tail, alphas_);
    if (mc_.cif_cumsum) launch_cif_scan_cumsum(stream_, alphas_, B, T1, plan_);
    else launch_cif_scan(stream_, alphas_, B, T1, mc_.cif_threshold, plan_);
    prof_end("cif_misc");
    export_plan(B);
  }
  // BiCIF timestamp head: ConvTranspose1d, LSTM and the excluded cif_output2 MatMul are float nodes of the int8 export too
  // (quantize_dynamic is run with op_types_to_quantize = ["MatMul"]): the f16 path's head, beside the decoder
  if (mc_.timestamp_head) start_timestamp_head(B, T);
  int32_t L = read_back_plan(B);
  if (l_hook_) L = l_hook_(L);
  last_.B = B; last_.L = L; last_.V = V; last_.T = T;
  last_.ids.assign((size_t)B * L, 0);
  if (L == 0) { join_ts(); return; }
  // ---- decoder
  const int Md = B * L;
  const int64_t Mdp = round_up(Md, 128) + 128;
  size_t o2 = 0;
  auto c2 = [&](size_t bytes) { size_t o = o2; o2 += round_up((int64_t)bytes, (int64_t)kAlignQ); return o; };
  const size_t o_xd = c2(Mdp * D * 4), o_hd = c2(Mdp * F * 4);
  const size_t o_td = c2(Mdp * D * 4), o_tn2 = c2(Mdp * D * 4), o_q = c2(Mdp * D * 2), o_cx = c2(Mdp * D * 2);
  const size_t o_kv = c2((size_t)Mp * 2 * D * 2), o_lg = c2((size_t)Mdp * ldV * 4), o_ids = c2((size_t)Md * 8);
  const size_t o_qh = c2((size_t)(Mp + 256) * round_up(D, 128)), o_qhr = c2((size_t)(Mp + 256) * 4), o_qhp = c2(64);
  ensure(ws_dec_, o2);
  char* b2 = (char*)ws_dec_.p;
  float* xd = (float*)(b2 + o_xd); float* hd = (float*)(b2 + o_hd);
  float* t32 = (float*)(b2 + o_td); float* tn32 = (float*)(b2 + o_tn2);
  half_t* qd16 = (half_t*)(b2 + o_q); half_t* cx16 = (half_t*)(b2 + o_cx); half_t* kv16 = (half_t*)(b2 + o_kv);
  logits_ = (float*)(b2 + o_lg); ids_dev_ = (int64_t*)(b2 + o_ids); logits_ld_ = ldV;
  prof_begin("cif_misc", 0);
  if (mc_.cif_cumsum) launch_cif_gather_cumsum(stream_, H32_, alphas_, B, T, D, T1, plan_, L, xd);
  else launch_cif_gather(stream_, H32_, B, T, D, T1, plan_, L, xd);
  prof_end("cif_misc");
  const bool bias_branch = mc_.seaco && n_hotwords_ > 0;
  float* e0 = nullptr;                                 // SeACo: the bias decoder also starts from the CIF embeds
  float* hid32 = nullptr;
  if (bias_branch) {
    ensure(ws_seaco_in_, (size_t)2 * Mdp * D * 4);
    e0 = (float*)ws_seaco_in_.p;
    hid32 = e0 + (size_t)Mdp * D;
    PF_HIP(hipMemcpyAsync(e0, xd, (size_t)Md * D * 4, hipMemcpyDeviceToDevice, stream_));
  }
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2) {
    qgemm("gemm_dec_ffn1", w1, true, xd, nullptr, D, Md, hd, F, nullptr, 0, nullptr, 0, nullptr, 0, true, 0, 1.f, &n1);
    qgemm("gemm_dec_ffn2", w2, false, hd, nullptr, F, Md, t32, D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f, &fn);
  };
  // K / V of the encoder memory: every layer's MatMul quantises the SAME tensor — one DynamicQuantizeLinear result here
  QAct qH;
  {
    const int kp = (int)round_up(D, 128);
    qH.a = (int8_t*)(b2 + o_qh); qH.rowsum = (int32_t*)(b2 + o_qhr); qH.params = (float*)(b2 + o_qhp);
    quantize_act(qH, kp, H32_, nullptr, D, M, D, nullptr);
  }
  for (size_t i = 0; i < dec_.size(); ++i) {
    const DecLayer& Lr = dec_[i];
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, t32, Md, D, Lr.norm2.g, Lr.norm2.b, nullptr, 0, tn32, D);
    prof_end("layernorm");
    prof_begin("fsmn", 0);
    launch_fsmn_dec(stream_, tn32, Lr.fsmn_wT, plan_.token_num, B, L, D, mc_.kernel, xd);
    prof_end("fsmn");
    qgemm("gemm_dec_q", Lr.q, true, xd, nullptr, D, Md, nullptr, 0, qd16, D, nullptr, 0, nullptr, 0, false, D, qscale, &Lr.norm3);
    if (lin_quantised(Lr.kv32))
      qgemm("gemm_dec_kv", Lr.kv32, true, nullptr, nullptr, D, (int)M, nullptr, 0, kv16, 2 * D, nullptr, 0, nullptr, 0, false, 0, 1.f, nullptr, &qH);
    else
      gemm("gemm_dec_kv", Lr.kv32, H16_, D, (int)M, nullptr, 0, kv16, 2 * D, nullptr, 0, nullptr, 0, false, 0, 1.f);
    AttnArgs a{};
    a.q = qd16; a.q_bstride = (int64_t)L * D; a.q_rstride = D;
    a.k = kv16; a.v = kv16 + D; a.k_bstride = a.v_bstride = (int64_t)T * 2 * D; a.k_rstride = a.v_rstride = 2 * D;
    a.o = cx16; a.o_bstride = (int64_t)L * D; a.o_rstride = D;
    a.B = B; a.H = mc_.heads; a.Lq = L; a.Lk = T;
    const bool cx_range = attention_reports_range(a);
    a.range = cx_range ? q_part_ : nullptr;
    prof_begin("attn_cross", 4.0 * B * (double)L * T * D);
    launch_attention(stream_, a);
    prof_end("attn_cross");
    qgemm("gemm_dec_out", Lr.out, true, nullptr, cx16, D, Md, xd, D, nullptr, 0, xd, D, nullptr, 0, false, 0, 1.f, nullptr, nullptr,
          cx_range ? kRangeIn : 0);
  }
  ffn_dec(dec_final_norm1_, dec_final_w1_, dec_final_ffn_norm_, dec_final_w2_);
  if (bias_branch) {                                   // the bias decoder's second input: the after_norm hidden itself
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, t32, Md, D, dec_after_.g, dec_after_.b, nullptr, 0, hid32, D);
    prof_end("layernorm");
    qgemm("gemm_vocab", dec_out_, true, hid32, nullptr, D, Md, logits_, ldV, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f);
  } else {
    qgemm("gemm_vocab", dec_out_, true, t32, nullptr, D, Md, logits_, ldV, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f, &dec_after_);
  }
  prof_begin("argmax", 0);
  launch_argmax(stream_, logits_, Md, V, ldV, want_logits ? 2 : 1, ids_dev_);
  prof_end("argmax");
  if (bias_branch) seaco_head(B, L, e0, hid32, want_logits);
  join_ts();
  PF_HIP(hipMemcpyAsync(last_.ids.data(), ids_dev_, (size_t)Md * 8, hipMemcpyDeviceToHost, stream_));
}

// ---- SeACo bias branch in math_mode 2 (reference default for the flagship model: model.int8.onnx + model_eb.int8.onnx,
// Examples/OfflineAliParaformerAsrRecognizer.cs:17-21).  The hot-word embedder (Embedding + LSTM) holds no MatMul with a
// constant operand outside the LSTM nodes and stays on the f16 path; the bias decoder is the ASR decoder's structure.
// K / V rows of bias_embed [NJ, D]: every layer's MatMul quantises the same tensor (the per-tensor range of the tiled
// [B, NJ, D] input equals that of one copy), so it is quantised once.
void Engine::seaco_kv_int8(const float* hw32, const half_t* hw16, int NJ, half_t* kv16, int ldkv) {
  const int D = mc_.d_model, kp = (int)round_up(D, 128);
  const int64_t rp = round_up(NJ, 256) + 256;
  ensure(ws_seaco_q_, (size_t)rp * kp + (size_t)rp * 4 + 1024);
  QAct q;
  char* b = (char*)ws_seaco_q_.p;
  q.a = (int8_t*)b; q.rowsum = (int32_t*)(b + (size_t)rp * kp); q.params = (float*)(b + (size_t)rp * kp + (size_t)rp * 4);
  quantize_act(q, kp, hw32, nullptr, D, NJ, D, nullptr);
  for (size_t i = 0; i < sdec_.size(); ++i) {
    const Lin& kv = sdec_[i].kv32;
    // a [NJ, 2D] slice of the [NJ, ns * 2D] buffer: f16-only results with a row stride of ldkv
    if (lin_quantised(kv))
      qgemm("gemm_seaco", kv, true, nullptr, nullptr, D, NJ, nullptr, 0, kv16 + i * 2 * D, ldkv, nullptr, 0, nullptr, 0, false, 0, 1.f, nullptr, &q);
    else
      gemm("gemm_seaco", kv, hw16, D, NJ, nullptr, 0, kv16 + i * 2 * D, ldkv, nullptr, 0, nullptr, 0, false, 0, 1.f);
  }
}

// ONE pass of the bias decoder over R = B * L rows (xs); leaves after_norm(x) in hid [R, D].  The graph runs the decoder
// twice — on the CIF embeds and on the ASR decoder hidden — and each run has its own DynamicQuantizeLinear nodes, i.e. its
// own per-tensor ranges: the f16 path's single pass over 2 * B * L rows would merge the two ranges, so this mode runs two.
void Engine::seaco_decoder_int8(int B, int L, int NJ, float* xs, float* h32, float* t32, float* tn32, half_t* q16, half_t* ctx16,
                                const half_t* kv16, int ldkv, const int32_t* tn2, float* hid) {
  const int D = mc_.d_model, Fs = mc_.seaco_ffn, R = B * L;
  const float qscale = 1.0f / std::sqrt((float)(D / mc_.heads));
  if (!q_part_) q_part_ = (float*)dalloc(quant_scratch_bytes());
  auto ffn_dec = [&](const LNp& n1, const Lin& w1, const LNp& fn, const Lin& w2) {
    qgemm("gemm_seaco", w1, true, xs, nullptr, D, R, h32, Fs, nullptr, 0, nullptr, 0, nullptr, 0, true, 0, 1.f, &n1);
    qgemm("gemm_seaco", w2, false, h32, nullptr, Fs, R, t32, D, nullptr, 0, nullptr, 0, nullptr, 0, false, 0, 1.f, &fn);
  };
  for (size_t i = 0; i < sdec_.size(); ++i) {
    const DecLayer& Lr = sdec_[i];
    ffn_dec(Lr.norm1, Lr.w1, Lr.ffn_norm, Lr.w2);
    prof_begin("layernorm", 0);
    launch_layernorm(stream_, t32, R, D, Lr.norm2.g, Lr.norm2.b, nullptr, 0, tn32, D);
    prof_end("layernorm");
    prof_begin("fsmn", 0);
    launch_fsmn_dec(stream_, tn32, Lr.fsmn_wT, tn2, B, L, D, mc_.seaco_kernel, xs);
    prof_end("fsmn");
    qgemm("gemm_seaco", Lr.q, true, xs, nullptr, D, R, nullptr, 0, q16, D, nullptr, 0, nullptr, 0, false, D, qscale, &Lr.norm3);
    AttnArgs a{};
    a.q = q16; a.q_bstride = (int64_t)L * D; a.q_rstride = D;
    a.k = kv16 + i * 2 * D; a.v = kv16 + i * 2 * D + D;
    a.k_bstride = a.v_bstride = 0; a.k_rstride = a.v_rstride = ldkv;
    a.o = ctx16; a.o_bstride = (int64_t)L * D; a.o_rstride = D;
    a.B = B; a.H = mc_.heads; a.Lq = L; a.Lk = NJ;
    const bool cx_range = attention_reports_range(a);
    a.range = cx_range ? q_part_ : nullptr;
    prof_begin("attn_seaco", 4.0 * B * (double)L * NJ * D);
    launch_attention(stream_, a);
    prof_end("attn_seaco");
    qgemm("gemm_seaco", Lr.out, true, nullptr, ctx16, D, R, xs, D, nullptr, 0, xs, D, nullptr, 0, false, 0, 1.f, nullptr, nullptr,
          cx_range ? kRangeIn : 0);
  }
  ffn_dec(seaco_final_norm1_, seaco_final_w1_, seaco_final_ffn_norm_, seaco_final_w2_);
  prof_begin("layernorm", 0);
  launch_layernorm(stream_, t32, R, D, seaco_after_.g, seaco_after_.b, nullptr, 0, hid, D);
  prof_end("layernorm");
}

// stand-alone operator (parity tests): one dynamically quantised Linear, nothing cached
void Engine::op_qlinear(const float* x, const float* W, const float* bias, int M, int N, int K, int relu, int x_is_f16, float* y,
                        uint8_t* xq_out, float* aparams_out, uint8_t* wq_out, float* wscale_out, int32_t* wzp_out) {
  PF_HIP(hipSetDevice(device_));
  PF_CHECK(x && W && y && M > 0 && N > 0 && K > 0 && K % 4 == 0, PF_ERR_INVALID_ARG, "op_qlinear: bad arguments (K must be a multiple of 4)");
  const int Kp = (int)round_up(K, 128), ldy = (int)round_up(N, 4);
  const int64_t Mp = round_up(M, 256) + 256, Np = round_up(N, 128);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up((int64_t)bytes, (int64_t)kAlignQ); return o; };
  const size_t ox = carve((size_t)M * K * 4), ox16 = carve((size_t)M * K * 2), oW = carve((size_t)N * K * 4), ob = carve((size_t)(N + 8) * 4);
  const size_t oa = carve((size_t)Mp * Kp), ors = carve((size_t)Mp * 4), opar = carve(64), osc = carve(quant_scratch_bytes());
  const size_t ow = carve((size_t)Np * Kp), ocs = carve((size_t)(N + 8) * 4), ozp = carve((size_t)(N + 8) * 4), ows = carve((size_t)(N + 8) * 4);
  const size_t oy = carve((size_t)Mp * ldy * 4);
  // bit 1 of x_is_f16: the f16-result kernel (deferred packed epilogue + the result's range for the next quantiser)
  const bool f16_result = (x_is_f16 & 2) != 0;
  x_is_f16 &= 1;
  const int ld16 = (int)round_up(N, 8), Kq = (int)round_up(N, 128);
  const size_t oy16 = carve((size_t)Mp * ld16 * 2), odz = carve((size_t)(N + 8) * 4), orng = carve(quant_scratch_bytes()), osc2 = carve(quant_scratch_bytes());
  const size_t oq2 = carve((size_t)Mp * Kq), ors2 = carve((size_t)Mp * 4), opa = carve(64), opb = carve(64);
  ensure(ws_tmp_, off);
  char* base = (char*)ws_tmp_.p;
  PF_HIP(hipMemcpyAsync(base + ox, x, (size_t)M * K * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemcpyAsync(base + oW, W, (size_t)N * K * 4, hipMemcpyHostToDevice, stream_));
  if (bias) PF_HIP(hipMemcpyAsync(base + ob, bias, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
  PF_HIP(hipMemsetAsync(base + ow, 0, (size_t)Np * Kp, stream_));
  PF_HIP(hipMemsetAsync(base + oa, 0, (size_t)Mp * Kp, stream_));
  launch_quantize_weight(stream_, (const float*)(base + oW), N, K, (int8_t*)(base + ow), Kp, (int32_t*)(base + ocs), (int32_t*)(base + ozp),
                         (float*)(base + ows));
  if (x_is_f16) {                                      // the activation arrives as f16 (attention context, FFN hidden): exact widening
    launch_f32_to_f16(stream_, (const float*)(base + ox), M, K, K, (half_t*)(base + ox16), K);
    launch_quantize_rows(stream_, nullptr, (const half_t*)(base + ox16), M, K, K, (int8_t*)(base + oa), Kp, (int32_t*)(base + ors),
                         (float*)(base + opar), (unsigned*)(base + osc));
  } else {
    launch_quantize_rows(stream_, (const float*)(base + ox), nullptr, M, K, K, (int8_t*)(base + oa), Kp, (int32_t*)(base + ors),
                         (float*)(base + opar), (unsigned*)(base + osc));
  }
  GemmI8Args g{};
  g.A = (int8_t*)(base + oa); g.lda = Kp; g.W = (int8_t*)(base + ow); g.ldw = Kp;
  g.rowsum = (int32_t*)(base + ors); g.colsum = (int32_t*)(base + ocs); g.wzp = (int32_t*)(base + ozp); g.wscale = (float*)(base + ows);
  g.aparams = (float*)(base + opar); g.bias = bias ? (const float*)(base + ob) : nullptr;
  g.M = M; g.N = N; g.K = K; g.Kpad = Kp; g.relu = relu;
  float pa[2] = {0, 0}, pb[2] = {1, 1};
  std::vector<uint16_t> y16;
  if (f16_result) {
    launch_pack_dz(stream_, g.colsum, g.wzp, N, K, (int32_t*)(base + odz));
    g.out_f16 = (half_t*)(base + oy16); g.ldc16 = ld16; g.dz = (int32_t*)(base + odz); g.out_padded = 1; g.range_out = (float*)(base + orng);
    launch_gemm_i8(stream_, g);
    PF_CHECK(std::strncmp(last_gemm_kernel(), "gemm_i8f", 8) == 0, PF_ERR_DEVICE, "op_qlinear: the f16-result kernel was not selected");
    // the range the kernel reports must be the range a min / max pass over its output finds: quantise the output both ways
    launch_quantize(stream_, nullptr, (const half_t*)(base + oy16), M, N, ld16, (int8_t*)(base + oq2), Kq, (int32_t*)(base + ors2),
                    (float*)(base + opa), (const float*)(base + orng));
    launch_quantize_rows(stream_, nullptr, (const half_t*)(base + oy16), M, N, ld16, (int8_t*)(base + oq2), Kq, (int32_t*)(base + ors2),
                         (float*)(base + opb), (unsigned*)(base + osc2));
    PF_HIP(hipMemcpyAsync(pa, base + opa, 8, hipMemcpyDeviceToHost, stream_));
    PF_HIP(hipMemcpyAsync(pb, base + opb, 8, hipMemcpyDeviceToHost, stream_));
    y16.resize((size_t)M * ld16);
    PF_HIP(hipMemcpyAsync(y16.data(), base + oy16, y16.size() * 2, hipMemcpyDeviceToHost, stream_));
  } else {
    g.out_f32 = (float*)(base + oy); g.ldc32 = ldy;
    launch_gemm_i8(stream_, g);
    PF_HIP(hipMemcpy2DAsync(y, (size_t)N * 4, base + oy, (size_t)ldy * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost, stream_));
  }
  std::vector<int8_t> aq, wq;
  std::vector<int32_t> zp((size_t)N);
  if (xq_out) { aq.resize((size_t)M * Kp); PF_HIP(hipMemcpyAsync(aq.data(), base + oa, aq.size(), hipMemcpyDeviceToHost, stream_)); }
  if (wq_out) { wq.resize((size_t)N * Kp); PF_HIP(hipMemcpyAsync(wq.data(), base + ow, wq.size(), hipMemcpyDeviceToHost, stream_)); }
  if (aparams_out) PF_HIP(hipMemcpyAsync(aparams_out, base + opar, 8, hipMemcpyDeviceToHost, stream_));
  if (wscale_out) PF_HIP(hipMemcpyAsync(wscale_out, base + ows, (size_t)N * 4, hipMemcpyDeviceToHost, stream_));
  if (wzp_out) PF_HIP(hipMemcpyAsync(zp.data(), base + ozp, (size_t)N * 4, hipMemcpyDeviceToHost, stream_));
  PF_HIP(hipStreamSynchronize(stream_));
  PF_CHECK(!f16_result || (pa[0] == pb[0] && pa[1] == pb[1]), PF_ERR_DEVICE,
           "op_qlinear: the range reported by the GEMM epilogue differs from a min / max pass over its output");
  if (f16_result)
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        half_t h;
        std::memcpy(&h, &y16[(size_t)m * ld16 + n], 2);
        y[(size_t)m * N + n] = (float)h;
      }
  if (xq_out)
    for (int m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k) xq_out[(size_t)m * K + k] = (uint8_t)((int)aq[(size_t)m * Kp + k] + 128);
  if (wq_out)
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) wq_out[(size_t)n * K + k] = (uint8_t)((int)wq[(size_t)n * Kp + k] + 128);
  if (wzp_out)
    for (int n = 0; n < N; ++n) wzp_out[n] = zp[(size_t)n] + 128;
}

}  // namespace pf
