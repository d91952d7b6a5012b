// engine.h — device engine: weights, workspace and the forward pipeline.
// Replaces OfflineModel (ORT session, AliParaformerAsr/OfflineModel.cs:35-70) and the numeric
// body of IOfflineProj.ModelProj (AliParaformerAsr/IOfflineProj.cs:38).
#pragma once
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "json.h"
#include "kernels.h"
#include "shards.h"

namespace pf {

struct ModelCfg {
  std::string kind = "paraformer";
  int feat_dim = 560, d_model = 512, heads = 4, ffn = 2048, enc_layers = 50, tp_layers = 0, kernel = 11;
  int dec_layers = 16, vocab = 8404;
  float cif_threshold = 1.0f, cif_tail = 0.45f, cif_smooth = 1.0f, cif_noise = 0.0f;
  int cif_l_order = 1, cif_r_order = 1;
  bool cif_cumsum = false;           // cif_variant = "cumsum": the prefix-sum export (cif_v1_export) instead of the loop
  bool timestamp_head = false, seaco = false, use_itn = false;
  float cif_smooth2 = 0.25f, cif_noise2 = 0.01f;
  int upsample = 3;
  int seaco_layers = 4, seaco_ffn = 1024, seaco_kernel = 21, seaco_lstm_layers = 2, seaco_nobias = 8377;
  int kind_id() const { return kind == "sensevoicesmall" ? 1 : (kind == "seacoparaformer" ? 2 : 0); }
};

struct FrontendCfg {
  int fs = 16000, n_mels = 80, lfr_m = 7, lfr_n = 6;
  bool snip_edges = false;
  float dither = 0.f;
  uint32_t dither_seed = 0;
  std::string window = "hamming";
};

struct Lin {          // y = x W^T + b ; W stored f16 [Npad][Kpad]
  half_t* w = nullptr;
  const float* bias = nullptr;
  int N = 0, K = 0, Kpad = 0;
  const float* w32 = nullptr;         // the fp32 tensor [N][K] as stored in the container (fp32 parity mode)
};
struct LNp { const float* g = nullptr; const float* b = nullptr; int D = 0; };
// a Linear as onnxruntime's quantize_dynamic stores it (math_mode 2): w' = w_q - 128 as signed bytes [Npad][Kpad], K-contiguous
struct QLin {
  int8_t* w = nullptr;
  int32_t* colsum = nullptr;      // [N] sum_k w'
  int32_t* wzp = nullptr;         // [N] w_zp - 128
  float* wscale = nullptr;        // [N]
  int32_t* dz = nullptr;          // [N] packed column word of the f16-result kernel (kernels.h GemmI8Args::dz)
  const float* bias = nullptr;
  int N = 0, K = 0, Kpad = 0;
};
// a quantised activation tensor: a' = a_q - 128 [rows, Kpad], row sums of a', {scale, zero point}
struct QAct { int8_t* a = nullptr; int32_t* rowsum = nullptr; float* params = nullptr; };
struct EncLayer {
  LNp norm1, norm2; Lin qkv, out, w1, w2; float* fsmn_wT = nullptr; int d_in = 512;
  half_t* qkv_p = nullptr; float* qkv_bias_p = nullptr;   // qkv weight rows / bias in the tile order of gemm_qkvp_kernel (null: not built)
  half_t* ffn_wt = nullptr;                               // W1 | W2 in the fragment order of ffn_fused_kernel (k_ffn.hip; null: not built)
  half_t* qkv_t = nullptr;                                // the [Q | K | V] weight in that order too (three 512-row images): the PREVIOUS layer's launch runs this projection
  half_t* out_wt = nullptr;                               // the attention out-projection weight in the same kernel's fragment order
};
struct DecLayer {
  LNp norm1, ffn_norm, norm2, norm3; Lin w1, w2, q, out, kv32; float* fsmn_wT = nullptr;   // kv32: fp32 pointers only
  half_t* ffn_img = nullptr;   // (W1, gamma_F (.) W2, b1, colsum, W2 beta_F) in the fragment order of the split FFN form (k_ffn.hip; null: not built)
  half_t* out_wt = nullptr;    // the cross-attention out-projection weight in that kernel's fragment order: the NEXT block's launch runs it
  half_t* q_wt = nullptr;      // the cross-attention query projection in the same fragment order (k_decmid.hip; null: not built)
};

struct DevBuf {       // grow-only device allocation
  void* p = nullptr;
  size_t bytes = 0;
};

class Engine {
 public:
  explicit Engine(const pf_engine_config& cfg);
  ~Engine();

  // ---- front-end ---------------------------------------------------------
  int num_lfr_frames(int64_t n_samples) const;
  int num_fbank_frames(int64_t n_samples) const;
  void fbank_host(const float* samples, int64_t n, std::vector<float>& out, int& t80);
  void frontend_host(const float* samples, int64_t n, std::vector<float>& feats, int& t_lfr);
  // the same on samples that already live on this device (OfflineStream keeps the audio of its AddSamples call there)
  void frontend_staged_one(std::vector<float>& feats, int& t_lfr);
  void frontend_from_device(const float* samples_dev, int64_t n, std::vector<float>& feats, int& t_lfr);
  // stage_audio without the copy: utterance b = n[b] floats at the DEVICE address samples_dev[b] (4-byte aligned, alive until
  // the run has been synchronised); the batched front-end of run_staged reads them where they are
  void stage_device_audio(const float* const* samples_dev, const int64_t* n, int B, int force_T = 0);

  // ---- forward -----------------------------------------------------------
  // speech_dev: [B,T,feat] fp32 already on device (padded + sentinel)
  void forward_device(const float* speech_dev, int B, int T, bool want_logits);
  void forward_feats_host(const float* speech, int B, int T, bool want_logits);
  void model_proj_host(const float* const* speech, const int32_t* n_floats, int B, bool want_logits);
  // force_T > 0: pad to at least force_T LFR frames (a shard of a larger batch pads to the GLOBAL maximum, PadHelper.cs:25)
  void stage_audio(const float* const* samples, const int64_t* n, int B, int force_T = 0);
  void run_staged(bool want_logits);
  void fetch(pf_batch_out* out);
  void fetch_ids_device(int64_t* ids_dev, int l_cap, int32_t* L_out);   // last forward's ids -> caller's device buffer [B, l_cap], -1 padded
  // Results are kept per CALLING THREAD: a forward entry point (pf_forward_feats / pf_model_proj / pf_recognize)
  // publishes its outcome into the caller's slot, and pf_fetch from the same thread reads that slot, so the
  // two-call protocol (learn L, then fetch into right-sized buffers) is safe with concurrent callers on one
  // engine.  The staged API (pf_stage_audio / pf_run_staged) is engine state and single-caller by contract.
  void publish_thread_result();
  void drop_thread_result();
  void sync();
  // multi-device groups (group.h): the decoder length becomes the maximum over all shards of the batch
  void set_l_hook(std::function<int(int)> h) { l_hook_ = std::move(h); }
  const HostBatchOut& last_result() const { return last_; }
  void copy_logits(HostBatchOut& r);                 // host copy of the last forward's log-probs into r.logits
  const int64_t* ids_device() const { return ids_dev_; }
  const int32_t* token_num_device() const { return plan_.token_num; }
  hipStream_t stream() const { return stream_; }
  // SeACo: hotword ids [n, 10] (PadList output, EmbedSeacoModel.cs:70-123) used by the following forwards;
  // n = 0 -> bias_embed [B,0,512]: the bias branch is skipped and the ASR log-probs are returned
  void set_hotwords(const int32_t* hw, int n);
  // ---- streaming seams (OnlineRecognizer.cs EncoderProj / DecoderProj) ------
  void online_encoder(const float* speech, int B, int Tc, float* enc_out, float* alphas_out);
  void online_decoder(const float* enc, int B, int Tc, const float* embeds, int L, const int32_t* embeds_len,
                      const float* caches_in, float* logits_out, int64_t* ids_out, float* caches_out);
  int dec_layers() const { return (int)dec_.size(); }

  // ---- stand-alone ops (parity tests) -------------------------------------
  void op_lfr_cmvn_pad(const float* const* fbank, const int32_t* t80, int B, int sentinel, float* out,
                       int64_t cap, int32_t* tmax);
  void op_argmax(const float* x, int64_t rows, int V, int64_t* ids);
  void op_gemm(const float* A, const float* W, const float* bias, int M, int N, int K, int epi, float* C);
  void op_gemm_ex(const pf_gemm_desc& d, const float* A, const float* W, float* C);
  void op_linear32(const float* x, const float* W, const float* bias, const float* resid, int M, int N, int K, bool relu, float* y);
  void op_ffn32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, int M, int D, int F, float* y);
  void op_gemm_rc(const pf_gemm_rc_desc& d, const float* A, const float* W, float* x_out, float* n16_out, float* n32_out);
  // DynamicQuantizeLinear(x) + MatMulInteger + rescale (+ bias) (+ ReLU) on the int8 MFMA; optional outputs: the uint8
  // activations [M, K], {a_scale, a_zp}, the uint8 weights [N, K] and their per-channel scale / zero point
  void op_qlinear(const float* x, const float* W, const float* bias, int M, int N, int K, int relu, int x_is_f16, float* y,
                  uint8_t* xq_out, float* aparams_out, uint8_t* wq_out, float* wscale_out, int32_t* wzp_out);
  void op_dec_ffn_fused(const pf_dec_ffn_desc& ds, float* t_out, float* n_out, float* x_out);
  void op_ffn_fused(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* resid,
                    const float* g, const float* be, int M, float* x_out, float* n16_out, const pf_attn_ffn_desc* op = nullptr);
  void op_ffn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* resid,
              int M, int D, int F, float* y);
  void op_fsmn_enc(const float* v, const float* w, int B, int T, int D, int k, float* y);
  void op_fsmn_dec(const float* tn, const float* w, const int32_t* token_num, int B, int L, int D, int k, float* x);
  void op_logsoftmax_argmax(const float* x, int64_t rows, int V, float* y, int64_t* ids);
  void op_layernorm(const float* x, const float* g, const float* b, int64_t rows, int D, float* y);
  void op_attention(const float* q, const float* k, const float* v, int B, int Lq, int Lk, int H, float* o);
  void op_qkv_attention(const float* x, const float* w, const float* bias, int B, int T, int K, float* q_out, float* k_out,
                        float* v_out, float* ctx_out);
  void op_fsmn(const float* v, const float* w, const float* mask, int B, int T, int D, int k, float* y);
  void op_cif(const float* H, const float* alphas, int B, int T, int D, float thr, int Lcap, float* E,
              int32_t* fire_count, int32_t* token_num, int32_t* L_out);
  void op_encoder(const float* speech, int B, int T, float* H);

  // ---- profiling -----------------------------------------------------------
  void profile_enable(bool on) { prof_on_ = on; }
  void profile_reset();
  void profile_select(const std::string& cls) { prof_only_ = cls; }
  bool profile_get(const std::string& cls, double* ms, int64_t* launches, double* flops_per_launch);
  std::string profile_kernel(const std::string& cls) const;   // GEMM classes: the kernel the launcher chose ("" otherwise)
  double last_flops() const { return last_flops_; }

  const ModelCfg& model() const { return mc_; }
  const FrontendCfg& frontend() const { return fc_; }
  const std::vector<float>& embed_table() const { return embed_host_; }   // SenseVoice [16,560]
  std::mutex& mutex() { return mu_; }
  int device() const { return device_; }
  // the batched front-end of run_staged (LFR + CMVN + pad in one kernel) computes what frontend_host + PadSequence compute
  bool staged_frontend_matches_host() const { return fc_.lfr_m * fc_.n_mels == mc_.feat_dim && (!cmvn_shift_ || cmvn_dim_ == mc_.feat_dim); }
  bool has_device_prompt() const { return sv_prompt_ != nullptr; }

 private:
  struct Tensor { const float* dev = nullptr; std::vector<int64_t> shape; int64_t numel = 0; bool u8 = false; };   // u8: dev points at bytes
  struct ProfClass { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double flops = 0; int64_t n = 0; std::string kernel; };

  void load_weights(const pf_engine_config& cfg);
  const Tensor& tensor(const std::string& name) const;          // float32 tensors only (a u8 tensor here is a format error)
  const Tensor* tensor_u8(const std::string& name) const;       // nullptr when absent
  bool has_tensor(const std::string& name) const { return tensors_.count(name) != 0; }
  Lin make_lin(const std::string& prefix, bool bias);
  half_t* make_dec_ffn_image(const Lin& w1, const LNp& fn, const Lin& w2);
  LNp make_ln(const std::string& prefix, int width);
  void release();
  float* make_fsmn_wT(const std::string& name, int K = 0);
  void* dalloc(size_t bytes);
  void ensure(DevBuf& b, size_t bytes);
  void build_pe(int T);
  void encoder(const float* speech_dev, int B, int T, bool pre_encoded = false);
  // the LayerNorm applied to the residual stream right after a layer's FFN-down, and where its results go
  struct EncNext { LNp ln; half_t* n16 = nullptr; float* n32 = nullptr; bool keep_x = true; const EncLayer* next = nullptr; };
  bool qkv_done_ = false;            // the previous layer's launch has already written this layer's Q | K (blocked) and V
  int v_pp_ = 0;                     // which of the two V buffers the current layer reads (the fused launch writes the other)
  // first: 0 = not the first layer, 1 = first (x sqrt(d) + position encoding fused into norm1), 2 = first, input already encoded
  void enc_layer(const EncLayer& L, int first, const float* speech_dev, int B, int T, const EncNext& nx);
  void predictor_and_decoder(int B, int T, bool want_logits);
  void sensevoice_head(int B, int T, bool want_logits);
  void forward_fp32(const float* speech_dev, int B, int T, bool want_logits);   // math_mode 1 (k_fp32.hip)
  // math_mode 2 (engine_int8.cpp): every Linear as DynamicQuantizeLinear + MatMulInteger on v_mfma_i32_32x32x32_i8
  void forward_int8(const float* speech_dev, int B, int T, bool want_logits);
  const QLin& qlin(const Lin& l, bool bias = true);
  const QLin& qlin_raw(const float* w32, const float* bias, int N, int K);
  // y = dequant(quant(x) w_q^T) + bias [* scale on the first scale_cols columns] [+ add2] [+ resid] [ReLU]; x fp32 [M, ldx] or f16
  void qgemm(const char* cls, const Lin& w, bool bias, const float* x32, const half_t* x16, int ldx, int M, float* out32, int ld32,
             half_t* out16, int ld16, const float* resid, int ldr, const float* add2, int ld2, bool relu, int scale_cols, float scale,
             const LNp* ln = nullptr, const QAct* pre = nullptr, int range = 0);   // range: 1 = leave the result's {min, max} pairs
                                                                                    // in q_part_, 2 = the input's are there already
  bool lin_quantised(const Lin& l) const;            // is this Linear a DynamicQuantizeLinear + MatMulInteger pair in the model file?
  void fgemm_in_int8(const char* cls, const Lin& l, bool bias, const float* x32, const half_t* x16, int ldx, int M, float* out32, int ld32,
                     half_t* out16, int ld16, const float* resid, int ldr, const float* add2, int ld2, bool relu, int scale_cols,
                     float scale, const LNp* ln, int range);
  void seaco_kv_int8(const float* hw32, const half_t* hw16, int NJ, half_t* kv16, int ldkv);
  void seaco_decoder_int8(int B, int L, int NJ, float* xs, float* h32, float* t32, float* tn32, half_t* q16, half_t* ctx16,
                          const half_t* kv16, int ldkv, const int32_t* tn2, float* hid);
  enum { kRangeOut = 1, kRangeIn = 2 };
  // quantise an activation tensor into `dst` (min / max pass, quantise pass); ln: of LayerNorm(x32), which is never stored
  void quantize_act(const QAct& dst, int kpad, const float* x32, const half_t* x16, int ldx, int64_t M, int K, const LNp* ln,
                    bool have_range = false);
  void timestamp_head_fp32(int B, int T);
  void seaco_head_fp32(int B, int L, const float* e0, const float* hid, bool want_logits);
  // fp32 LSTM over rows [Bn * Tn] of x (row b * Tn + t): hout[(b * Tn + t) * ldh + col0 .. + D); reverse = time runs backwards
  void lstm_fp32(const float* x, int Bn, int Tn, const float* w_ih, const float* w_hh, const float* bias, bool reverse,
                 float* xg, float* gates, float* hbuf, float* cbuf, float* hout, int ldh, int col0);
  void enc_layer_fp32(const EncLayer& L, bool first, const float* speech_dev, int B, int T, float** bufs);
  void timestamp_head(int B, int T);
  void start_timestamp_head(int B, int T);
  void seaco_head(int B, int L, const float* e0, const float* hid32, bool want_logits);
  void gemm(const char* cls, const Lin& w, const half_t* A, int lda, int M, float* out32, int ld32,
            half_t* out16, int ld16, const float* resid, int ldr, const float* add2, int ld2, bool relu,
            int scale_cols, float scale, bool bias = true, int blocked = 0);   // blocked: 1 = out_f16 blocked, 2 = A blocked
  void gemm_small_call(const char* cls, const Lin& w, GemmSmallArgs g, bool bias = true);
  void dec_ffn_hidden(const char* cls, const Lin& w1, const LNp& fn, const half_t* xn16, int lda, int rows, float* h32, half_t* h16);
  void prof_begin(const char* cls, double flops);
  void prof_end(const char* cls);

  // dither stream: seed of call c = hash(dither_seed, c); an engine built with the same seed replays the same
  // features for the same sequence of front-end calls
  uint32_t dither_calls_ = 0;
  uint32_t next_dither_seed() { return fc_.dither_seed * 2654435761u + (dither_calls_++) * 40503u; }
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  int32_t* plan_host_ = nullptr; int32_t* plan_host_dev_ = nullptr; int plan_host_cap_ = 0;   // CIF counts exported into pinned host memory
  void ensure_plan_host(int B);
  void export_plan(int B);
  int32_t read_back_plan(int B);
  hipStream_t aux_stream_ = nullptr;   // carries the decoder-length read-back, so stream_ keeps running (K/V projections) meanwhile
  hipEvent_t ev_scan_ = nullptr;       // CIF scan finished
  // the BiCIF timestamp head depends on the encoder output and token_num only: it runs on its own stream beside the
  // decoder (its recurrence is 1500 dependent steps on 128 workgroups — latency, not throughput) and is joined before
  // the call's last copy (PF_TS_STREAM=0 keeps it on the main stream)
  hipStream_t ts_stream_ = nullptr;
  hipEvent_t ev_ts_ = nullptr, ev_enc_ = nullptr;
  bool ts_pending_ = false, ts_defer_copy_ = false;
  size_t ts_copy_floats_ = 0;
  void join_ts();
  bool no_rc_ = false, rc_ffn2_ = true, lstm_steps_ = false;
  bool qkv_tail_ = true;             // PF_QKV_TAIL: the next layer's Q | K | V projection behind the fused block, same launch
  bool attn_ffn_ = true;             // PF_ATTN_FFN: out-projection + FSMN + norm2 in front of the fused FFN block, one launch
  bool ffn_fused_ = true;            // PF_FFN_FUSED: the encoder FFN block as one launch (k_ffn.hip)
  void own_hardware_queue_ts();
  void own_hardware_queue();         // round 6: the main stream must not share a hardware queue with another live engine's (see engine.cpp)
  bool dec_mid_ = true;              // PF_DEC_MID: finishing pass + norm2 + FSMN + residual + norm3 + q-projection in one launch (k_decmid.hip)
  bool dec_out_chain_ = true;        // PF_DEC_OUT_CHAIN: a decoder layer's out-projection + the next norm1 in front of the next FFN launch
  bool dec_ffn_fused_ = true;        // PF_DEC_FFN: the decoder's FFN block (with its LayerNorm over the hidden) as the split form of the same kernel
  int ffn_fused_min_rows_ = 1200;    // PF_FFN_MIN: below, 64-row tiles leave most CUs idle and the persistent kernels tie or win (tools/mid_rows.py)
  bool no_small_fuse_ = false;
  bool dec_h32_ = false;             // PF_DEC_H32=1: decoder FFN hidden through fp32 (A/B switch)
  int dec_fuse_ = 1;                 // bit 1: FSMN + norm3, bit 2: out-projection + next norm1, bit 4: FFN-down + norm2 (row-complete GEMM)
  bool qkv_split_ = true;            // PF_QKV_SPLIT=0: the row-major 256 x 128 kernel for Q | K | V
  int qkv_split_min_tiles_ = 256;    // PF_QKV_MIN: least number of 256 x 192 tiles for which the split form is chosen
  int qkv_split_min_fill_ = 60;      // PF_QKV_FILL: ... and least fill (percent) of its rounds of tiles (85 was the break-even with ONE step in
                                     // flight; with two, the other engine's kernels run on the CUs a short last round leaves)
  int cus_ = 256;                    // compute units the persistent kernels size their grids for (cu_limit)
  unsigned* lstm_err_ = nullptr;     // device time-out word of the last persistent LSTM launch (checked at the next result sync)
  void check_async_errors();         // after a stream sync: raises what a kernel of the finished forward reported through a flag word
  int x3_attn_ = 0;                  // PF_X3_ATTN (math_mode 3): 0 = fp32-MFMA attention, 1 = x3 operands throughout, 2 = fp32 scores + x3 P V
  void attention32(const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs, const float* v, int64_t v_bs, int v_rs,
                   float* o, int64_t o_bs, int o_rs, int B, int H, int Lq, int Lk, bool only_operand = false);
  // LayerNorm of the fp32 graph whose result is ONLY the A operand of gemm32 calls that follow (xn names it): in math_mode 3
  // (D = 512, above the short-input threshold) it is written as that operand pair and xn is not touched
  void layernorm32(const float* x, int M, int D, const LNp& ln, float* xn);
  bool x3_one_ = true;               // PF_X3_ONE=0: an x3 Linear as two launches with an [M, N] fp32 intermediate (round 5's first form; A/B)
  bool x3_fuse_ = true;              // PF_X3_FUSE=0: LayerNorm / attention write fp32 and gemm32 splits (A/B)
  bool x3a_pair_only_ = false;       // x3a_src_ exists ONLY as the pair in ws_x3a_ (its fp32 form was never written)
  bool x3_mode_ = false;             // math_mode 3: the fp32 graph with every large Linear as three f16 MFMA products of (hi, lo') operand pairs
  void x3_forget(const float* W);          // drops W's cached pair image (stand-alone ops: their weights live in a scratch arena)
  std::map<std::pair<const float*, int>, half_t*> x3w_;   // (fp32 weight, rows N) -> its [lo' | hi] f16 pair image (built on first use; the fused
                                                          // Q | K | V product and the three separate ones name the same first row)
  DevBuf ws_x3a_, ws_x3t_, ws_x3h_;
  bool x3_pair_live_ = false; int x3_pair_M_ = 0, x3_pair_K_ = 0;   // ws_x3h_ holds the (hi | lo') pair the next gemm32 consumes
  enum { kX3OutPair = 1, kX3InPair = 2, kX3SameInput = 4 };
  const float* x3a_src_ = nullptr; int x3a_M_ = 0, x3a_K_ = 0, x3a_ld_ = 0; half_t* x3a_buf_ = nullptr;   // what ws_x3a_ holds the pair of
  // profile class of the gemm32 / attention32 calls that follow (bench.py's `exact` roofline; "gemm32_*" / "attn32_*")
  const char* cls32_ = "gemm32_misc";
  void gemm32_impl(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K, float* out, int ldc,
                   const float* resid, int ldr, bool relu, int scale_cols, float scale, int flags, const float* resid2);
  void gemm32(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K, float* out, int ldc,
              const float* resid, int ldr, bool relu, int scale_cols, float scale, int flags = 0, const float* resid2 = nullptr);
  bool fp32_mode_ = false;           // math_mode 1: every GEMM / attention product on the fp32 MFMA path (parity runs)
  bool int8_mode_ = false;           // math_mode 2: Linear layers dynamically quantised to uint8, products on the int8 MFMA
  std::map<const float*, QLin> qlins_;
  std::map<const float*, std::string> lin_names_;     // float32 `<linear>.weight` device pointer -> `<linear>` (stored int8 bytes lookup)
  std::vector<std::string> int8_exclude_;   // config key `int8_exclude`: Linear name prefixes a synthetic container keeps in float
  bool any_stored_q_ = false;        // the container carries stored bytes of an int8 export
  DevBuf ws_qf_, ws_seaco_q_;        // f16 operand of an un-quantised Linear; quantised bias_embed rows
  DevBuf ws_q_;                      // quantised activations: a' [Mp, Kpad], row sums, {scale, zp}, min / max scratch
  int8_t* q_a_ = nullptr; int32_t* q_rowsum_ = nullptr; float* q_params_ = nullptr; unsigned* q_scratch_ = nullptr;
  int64_t q_rows_ = 0; int q_kpad_ = 0;
  float* q_part_ = nullptr;          // {min, max} per workgroup of a min / max pass (k_quant.hip)
  void ensure_q(int64_t rows, int kpad);
  ModelCfg mc_;
  FrontendCfg fc_;
  std::mutex mu_;

  // weights
  void* blob_dev_ = nullptr;        // device copy of the PFW data section (owned unless external)
  bool blob_owned_ = false;
  std::map<std::string, Tensor> tensors_;
  std::vector<void*> owned_;        // every other device allocation
  std::vector<EncLayer> enc_, tp_;
  std::vector<DecLayer> dec_;
  LNp enc_after_, tp_norm_, dec_final_norm1_, dec_final_ffn_norm_, dec_after_;
  Lin cif_conv_, dec_kv_all_, dec_final_w1_, dec_final_w2_, dec_out_, ctc_;
  half_t* dec_final_img_ = nullptr;  // the final block's image for the split FFN form (DecLayer::ffn_img)
  const float* cif_out_w_ = nullptr;
  const float* cif_out_b_ = nullptr;
  struct LstmLayer { Lin ih; half_t* whh = nullptr; };
  std::vector<DecLayer> sdec_;      // SeACo bias decoder
  std::vector<LstmLayer> seaco_lstm_;
  Lin seaco_kv_all_, seaco_final_w1_, seaco_final_w2_, seaco_out_;
  LNp seaco_final_norm1_, seaco_final_ffn_norm_, seaco_after_;
  const float* seaco_embed_w_ = nullptr;
  std::vector<int32_t> hotwords_;   // [n_hotwords_, 10]
  int n_hotwords_ = 0;
  Lin ts_up_, ts_ih_;               // BiCIF: ConvTranspose1d as [3D, D], W_ih of both directions [8D, D]
  half_t* ts_whh_ = nullptr;        // [2][4D][D]
  half_t* ts_whh_x3_ = nullptr;     // math_mode 3: [2][4D][hi (D) | lo' (D)] pair rows of the fp32 W_hh (built at first use)
  const float* ts_out_w_ = nullptr;
  const float* ts_out_b_ = nullptr;
  std::vector<float> embed_host_;
  float* sv_prompt_ = nullptr;      // SenseVoice: device [4, feat_dim] query rows (effective ids)

  // front-end
  FbankTables* fb_ = nullptr;
  float* cmvn_shift_ = nullptr;
  float* cmvn_scale_ = nullptr;
  int cmvn_dim_ = 0;

  // workspace
  DevBuf ws_f32_;
  float* small_ws_ = nullptr;        // short-input GEMM: split partials
  float* cif_conv_w32_ = nullptr;    // fp32 mode: the CIF conv as a [D][taps*D] GEMM operand
  float* ts_up_w32_ = nullptr;       // fp32 mode: the transposed conv as a [(j, out)][in] GEMM operand
  DevBuf ws_audio_, ws_meta_, ws_fbank_, ws_speech_, ws_enc_, ws_dec_, ws_decffn_, ws_kv_, ws_pe_, ws_tmp_, ws_ts_, ws_seaco_, ws_seaco_in_, ws_seaco_hw_;
  bool seaco_hw_valid_ = false;     // ws_seaco_hw_ holds the embedder output / K,V rows of the CURRENT hot-word list (f16 path)
  int pe_T_ = 0;
  // encoder views (valid after encoder())
  float* x_ = nullptr; half_t* xn16_ = nullptr; half_t* qkv16_ = nullptr; half_t* ctx16_ = nullptr;
  float* fsm_ = nullptr; half_t* h16_ = nullptr; float* H32_ = nullptr; half_t* H16_ = nullptr;
  float* alphas_ = nullptr; CifPlan plan_{};
  float* us_peak_ = nullptr;
  struct { const float* xg = nullptr; int B = 0, T3 = 0; } lstm_graph_key_;
  hipGraphExec_t lstm_graph_exec_ = nullptr;
  // decoder views
  float* logits_ = nullptr; int64_t* ids_dev_ = nullptr; int logits_ld_ = 0;
  // staged audio
  std::vector<int64_t> st_n_; std::vector<int32_t> st_t80_; int st_B_ = 0, st_T_ = 0;
  int64_t st_total_frames_ = 0;
  const float* st_audio_ext_ = nullptr;   // base of the externally staged audio (stage_device_audio); null: ws_audio_
  HostBatchOut last_;
  uint64_t uid_ = 0;                 // key of this engine in the per-thread result store
  static uint64_t register_uid();
  static void unregister_uid(uint64_t id);
  bool last_logits_ = false;
  double last_flops_ = 0;

  std::function<int(int)> l_hook_;
  bool prof_on_ = false;
  std::string prof_only_;
  std::map<std::string, ProfClass> prof_;
};

}  // namespace pf
