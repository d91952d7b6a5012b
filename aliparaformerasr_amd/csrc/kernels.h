// kernels.h — launchers of the gfx950 kernels (implemented in k_*.hip)
#pragma once
#include "common.h"

namespace pf {

// ---------------------------------------------------------------- GEMM ------
// C[M,N] = A[M,K] * W[N,K]^T (+ bias) ; f16 operands, f32 accumulate (MFMA 32x32x16).
// A rows must be readable up to round_up(M,128), W rows up to round_up(N,128); K is the
// padded depth (multiple of 64; pad columns of A and W hold zeros).
struct GemmArgs {
  const half_t* A; int lda;
  const half_t* W; int ldw;
  const float* bias;          // [N] or null
  int M, N, K;
  // epilogue
  float* out_f32; int ldc32;  // optional fp32 result
  half_t* out_f16; int ldc16; // optional f16 result
  const float* resid; int ldr;   // += resid[m,n] (fp32), may alias out_f32
  const float* add2; int ld2;    // += add2[m,n]  (fp32), e.g. the FSMN memory
  int relu;                      // max(v,0) after bias/resid
  int scale_cols; float scale;   // columns n < scale_cols are multiplied by scale (after bias); multiple of 64
  int out_padded;                // out_f16 has >= round_up(M,256) writable rows (enables whole-tile stores)
  // "blocked" f16 activation layout, element (m, n) of an [M, N] matrix at
  //   ((m/32 * N/8 + n/8) * 32 + m%32) * 8 + n%8      (32 rows x 8 columns = 512 contiguous bytes)
  // It is what a D^T MFMA fragment stores as whole lines WITHOUT a transposition and what the A-operand
  // LDS-DMA reads as 1 KiB contiguous pieces that land conflict-free (no swizzle).  Used for the FFN hidden:
  // FFN-up writes it (out_blocked, f16-only results, N % 64 == 0), FFN-down reads it (a_blocked, lda ignored).
  int out_blocked;
  int a_blocked;
  float* small_ws;               // scratch of the short-input kernel (k_gemm_small.hip, gemm_small_ws_bytes() bytes, one per
                                 // stream); null = never dispatch to it
  int f16_lo_off;                // > 0 (selects the fp32-kind epilogue): out_f16 receives the result as the operand pair of an
                                 // "x3" product (math_mode 3): hi = f16(v) at column n, lo' = f16((v - hi) * 2^11) at n + f16_lo_off
  // K-loop wrap (math_mode 3: the three partial products of an x3 Linear in ONE accumulation).  k_wrap > 0: after k_wrap
  // k-steps (of 64 columns) the accumulators are multiplied by wrap_scale and the operand cursors step BACK by a_wrap / w_wrap
  // columns, then the loop runs on to K / 64 steps in total:  A = [hi_x | lo'_x], W = [lo'_W | hi_W] (both 2 Kp wide),
  // K = 3 Kp, k_wrap = 2 Kp / 64, a_wrap = 2 Kp, w_wrap = Kp, wrap_scale = 2^-11  gives
  //   (hi_x lo'_W^T + lo'_x hi_W^T) 2^-11 + hi_x hi_W^T   — the small terms first, one fp32 accumulator, no intermediate in HBM.
  int k_wrap, a_wrap, w_wrap; float wrap_scale;
  int force_mi;                  // 0 = choose by shape; 1 / 2 = 128- / 256-row tiles of gemm_f16_pp3, 4 = the short-input kernel, 5 = the persistent 256 x 256 kernel (blocked result)
                                 // (stand-alone op tests; 4 / 5 fail when that kernel does not apply)
};
void launch_gemm(hipStream_t s, const GemmArgs& a);
// name of the kernel the calling thread's most recent GEMM launcher chose (bench.py prints it next to the class it
// reports as `roofline`; the same name appears in the rocprofv3 kernel trace)
const char* last_gemm_kernel();
void note_gemm_kernel(const char* name);
// persistent 256 x 192 tile kernel for the encoder's fused Q | K | V projection (k_gemm_qkv.hip): Q (scaled) and K leave in
// the blocked layout as one [Mpad, 1024] matrix, V row-major [Mpad, ldv].  Wp / bias_p: the weight rows and bias in tile order
// (launch_qkv_permute: [1536, ldw] f16 from the [Q | K | V] weight, bias may be null -> zeros); out rows up to round_up(M, 256).
bool gemm_qkvp_applicable(int M, int K, int lda, int ldw, int ldv);
void launch_qkv_permute(hipStream_t s, const half_t* w, int ldw, const float* bias, half_t* wp, float* bp);
void launch_gemm_qkvp(hipStream_t s, const half_t* A, int lda, const half_t* Wp, int ldw, const float* bias_p, int M, int K,
                      float qscale, half_t* out_qk, half_t* out_v, int ldv);
// persistent 256 x 256 tile kernel for the blocked-layout result (FFN-up; k_gemm_big.hip)
bool gemm_bigp_applicable(const GemmArgs& a);
void launch_gemm_bigp(hipStream_t s, const GemmArgs& a, int cus);

// ---- int8 (u8 x u8 dynamic quantisation as onnxruntime's quantize_dynamic emits it; k_gemm.hip gemm_i8_pp3 + k_quant.hip)
//   y[m,n] = float( sum_k (a_q[m,k] - a_zp)(w_q[n,k] - w_zp[n]) ) * (a_scale * w_scale[n]) + bias[n]  [+ add2 + resid] [ReLU]
// The device keeps SIGNED bytes a' = a_q - 128, w' = w_q - 128 (the matrix cores multiply signed int8) plus the row /
// column sums that turn sum a' w' back into the exact integer above.
struct GemmI8Args {
  const int8_t* A; int lda;            // [M, Kpad] a', rows readable up to round_up(M, 256); lda (bytes) % 16 == 0
  const int8_t* W; int ldw;            // [round_up(N,128), Kpad] w', K-contiguous; pad columns of A and W hold 0
  const int32_t* rowsum;               // [M] sum_k a'
  const int32_t* colsum;               // [N] sum_k w'
  const int32_t* wzp;                  // [N] w_zp - 128
  const float* wscale;                 // [N]
  const float* aparams;                // device [2]: a_scale, a_zp (written by launch_quantize_rows)
  const float* bias;                   // [N] or null
  int M, N, K, Kpad;                   // K = true depth, Kpad = multiple of 128
  float* out_f32; int ldc32; half_t* out_f16; int ldc16;
  const float* resid; int ldr; const float* add2; int ld2;
  int relu; int scale_cols; float scale;
  // f16-only results (out_f16, nothing else) into a buffer whose rows are padded to the tile height take the deferred
  // packed epilogue when dz is given: dz[n] = (colsum[n] - K wzp[n]) * 256 + (wzp[n] & 255)  (launch_pack_dz)
  const int32_t* dz; int out_padded;
  float* range_out;                    // null, or quant_scratch_bytes(): {min, max} of the results per workgroup (pass 1 of the consumer's quantiser)
};
void launch_pack_dz(hipStream_t s, const int32_t* colsum, const int32_t* wzp, int N, int K, int32_t* dz);
void launch_gemm_i8(hipStream_t s, const GemmI8Args& a);
// DynamicQuantizeLinear over the WHOLE tensor x [rows, cols] (fp32, or f16 when x16 is given): min / max (including 0)
// -> a_scale = (max - min) / 255, a_zp = rne(clamp(-min / a_scale, 0, 255)); q = clamp(rne(x / a_scale) + a_zp, 0, 255);
// writes a' = q - 128 into out [rows, ld] (pad columns 0), the row sums of a', and {a_scale, a_zp} into params[2].
// scratch: >= 16 bytes of device memory (min / max accumulators).
void launch_quantize_rows(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, int8_t* out, int ld,
                          int32_t* rowsum, float* params, unsigned* scratch);
// the two passes as separate launches; part: quant_scratch_bytes() of scratch ({min, max} per workgroup of pass 1)
size_t quant_scratch_bytes();
void launch_minmax(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, float* part);
void launch_quantize(hipStream_t s, const float* x32, const half_t* x16, int64_t rows, int cols, int ldx, int8_t* out, int ld,
                     int32_t* rowsum, float* params, const float* part);
// LayerNorm(x) quantised without ever storing LayerNorm(x): pass 1 min / max, pass 2 quantise (both normalise in registers)
void launch_ln_minmax(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma, const float* beta, float* part);
void launch_ln_quantize(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma, const float* beta, int8_t* out, int ld,
                        int32_t* rowsum, float* params, const float* part);
// per-output-channel weight quantisation as onnxruntime.quantization quantize_dynamic(per_channel=True, weight_type=QUInt8)
// does it: row n of W [N, K] (fp32): rmin = min(0, min), rmax = max(0, max), scale = (rmax - rmin) / 255,
// zp = rne(-rmin / scale) clamped to [0, 255], q = clamp(rne(w / scale) + zp, 0, 255) -> w' = q - 128 [N, ld], colsum, wzp = zp - 128, wscale
void launch_quantize_weight(hipStream_t s, const float* W, int N, int K, int8_t* out, int ld, int32_t* colsum, int32_t* wzp, float* wscale);
void launch_import_weight(hipStream_t s, const uint8_t* Q, const uint8_t* zp, const float* scale, int N, int K, int8_t* out, int ld,
                          int32_t* colsum, int32_t* wzp, float* wscale);

// Row-complete GEMM for N = 512 (k_gemm_rc.hip): x = resid + A W^T + bias + FSMN(V); n = LayerNorm(x).
// One workgroup = 64 complete rows, so the residual add, the FSMN memory and the following LayerNorm are its epilogue.
struct GemmRcArgs {
  const half_t* A; int lda; int a_blocked;       // [M,K] f16 (row-major, or the blocked activation layout)
  const half_t* W; int ldw;                      // [512, K] f16, K-contiguous
  const float* bias;                             // [512] or null
  int M, K;
  const float* resid; int ldr;                   // fp32 [M,512] or null; may alias out_x
  float* out_x; int ldx;                         // fp32 [M,512] result x, or null when only LayerNorm(x) is kept
  const half_t* fsmn_v; int ldv;                 // f16 V slice [M, ldv] (null = no FSMN term)
  const float* fsmn_wT; int fsmn_k; int T;       // taps [k][512]; utterances are runs of T rows
  const float* ln_g; const float* ln_b; float eps;   // LayerNorm over the 512 columns (null = none)
  half_t* out_n16; int ldn16;                    // f16 LayerNorm result (next GEMM's operand) or null
  float* out_n32; int ldn32;                     // fp32 LayerNorm result or null
};
// short inputs (k_gemm_small.hip): one-shot bricks, K > 576 as split partials + a row-wise reduction that also takes
// the LayerNorm behind the GEMM; optional FSMN-memory epilogue (attention out-projection)
struct GemmSmallArgs {
  const half_t* A; int lda;                      // f16 operand [M,K]
  const half_t* W; int ldw; const float* bias;   // [N,K] f16, K-contiguous
  int M, N, K;
  float* out_f32; int ldc32; half_t* out_f16; int ldc16;
  const float* resid; int ldr; const float* add2; int ld2;           // fp32 [M,N]; resid may alias out_f32
  int relu; int scale_cols; float scale;
  const half_t* fsmn_v; int ldv; const float* fsmn_wT; int fsmn_k; int T;   // + FSMN memory of the V slice (k = 11, N = 512, K <= 576)
  const float* post_ln_g; const float* post_ln_b;                    // K > 576 only: LayerNorm of the result ->
  half_t* post_n16; int ldn16; float* post_n32; int ldn32;           //   f16 / fp32 (out_f32 / out_f16 stay optional)
  float* ws;                                     // gemm_small_ws_bytes() of scratch (K > 576), one per stream
};
size_t gemm_small_ws_bytes();
int gemm_small_max_rows();                       // rows up to which the pipeline uses this kernel (PF_SMALL_M)
bool gemm_small_applicable(const GemmSmallArgs& a);
void launch_gemm_small(hipStream_t s, const GemmSmallArgs& a);
void launch_gemm_rc(hipStream_t s, const GemmRcArgs& a);
// The encoder's whole feed-forward block in one launch (k_ffn.hip): x = resid + relu(A W1^T + b1) W2^T + b2;
// n = LayerNorm(x).  d_model 512, hidden 2048; 64-row tiles, the hidden stays in LDS, W1 / W2 are streamed from their
// fragment-ordered images (launch_ffn_retile, once per layer at load) straight into registers.
struct FfnFusedArgs {
  const half_t* A; int lda;                      // [M,512] f16 (LayerNorm norm2 of the residual stream), rows readable up to round_up(M,64)
  const half_t* Wt;                              // ffn_fused_weight_bytes(): W1t | W2t
  const float* b1; const float* b2;              // [2048], [512]
  int M;
  const float* resid; int ldr;                   // fp32 [M,512] or null; may alias out_x
  float* out_x; int ldx;                         // fp32 [M,512] or null
  const float* ln_g; const float* ln_b; float eps;
  half_t* out_n16; int ldn16; float* out_n32; int ldn32;
  // ctx != null: the attention out-projection runs in front of the block in the same launch (A is ignored):
  //   x_mid = resid + ctx Wo^T + bo + FSMN(V) (kept on chip);  the block's operand = LayerNorm(x_mid; ln2) stays in LDS;  x = x_mid + FFN(.)
  const half_t* ctx; int lda_c;                  // attention context [M,512] f16
  const half_t* Wot; const float* bo;            // launch_ffn_retile_out image of Wo [512,512]; bias [512]
  const half_t* fsmn_v; int ldv; const float* fsmn_wT; int T;   // V slice (f16, row stride ldv), taps [11][512], utterances = runs of T rows (T >= 8)
  const float* ln2_g; const float* ln2_b;        // norm2
  // (resid may be null: the first layer)
  // Wqt != null (needs ctx and ln_g): the NEXT layer's fused Q | K | V projection of LayerNorm(x; ln_g, ln_b) behind the block,
  // same launch: Wqt = three launch_ffn_retile_out images ([Q | K | V] weight rows 0-511, 512-1023, 1024-1535), bq [1536];
  // Q (x qscale) and K -> blocked [Mpad, 1024] out_qk, V -> row-major out_v (row stride ldvo); out_n16 / out_n32 stay optional
  const half_t* Wqt; const float* bq; half_t* out_qk; half_t* out_v; int ldvo; float qscale;
};
bool ffn_fused_applicable(int D, int F);
size_t ffn_outproj_weight_bytes();
void launch_ffn_retile_out(hipStream_t s, const half_t* Wo, int ldw, half_t* Wot);
size_t ffn_fused_weight_bytes();
void launch_ffn_retile(hipStream_t s, const half_t* W1, int ldw1, const half_t* W2, int ldw2, half_t* Wt);
void launch_ffn_fused(hipStream_t s, const FfnFusedArgs& a);

// The decoder's position-wise block  t = LN_F(relu(A W1^T + b1)) W2^T  [, n = LayerNorm(t; ln_g, ln_b)]  (k_ffn.hip, split form):
// the same chunked kernel, the hidden range of a 64-row tile shared by 1 | 2 | 3 | 4 | 8 workgroups (ffn_dec_splits: the decoder
// has few rows), LN_F applied afterwards from row statistics collected on the way.  D = 512, F = 2048 only.
struct FfnDecArgs {
  const half_t* A; int lda;                      // norm1(x) as f16 [M,512] (ignored when ctx != null)
  // ctx != null: the PREVIOUS layer's cross-attention out-projection in front of the block, same launch:
  //   x = resid + ctx Wo^T + bo -> out_x (fp32, must NOT alias resid);  the block's operand = LayerNorm(x; ln1_g, ln1_b), kept in LDS
  const half_t* ctx; int lda_c; const half_t* Wot; const float* bo;      // Wot: launch_ffn_retile_out image of Wo [512,512]
  const float* resid; int ldr; float* out_x; int ldx;
  const float* ln1_g; const float* ln1_b; float eps1;
  const half_t* img;                             // launch_ffn_dec_retile image of (W1, gamma_F (.) W2, b1, colsum, W2 beta_F)
  void* ws;                                      // ffn_dec_workspace_bytes(M)
  int M;
  int splits;                                    // 0 = ffn_dec_splits(M); 1 | 2 | 3 | 4 | 8 forces the form (ws must then hold that many)
  float eps_hidden;                              // epsilon of LN_F
  float* t32; int ldt;                           // t, fp32 [M,512] or null
  const float* ln_g; const float* ln_b; float eps;
  float* n32; int ldn32; half_t* n16; int ldn16; // LayerNorm(t) or null
  bool no_finish = false;                        // the shares stay in ws: launch_dec_mid (k_decmid.hip) finishes them
};
const float* ffn_dec_image_cd(const half_t* img);  // the (c | d) vectors of the finishing pass inside the image

// the middle of a decoder layer in one launch (k_decmid.hip, round 6): finishing pass of the split FFN + norm2 + FSMN memory +
// residual + norm3 + q-projection, on the workspace a launch_ffn_dec(no_finish) left behind
struct DecMidArgs {
  const void* ws; const half_t* img; int splits; float eps_hidden;   // as the FfnDecArgs of the split launch in front
  const float* n2_g; const float* n2_b; float eps2;                   // norm2
  const float* fsmn_wT; int k; const int32_t* token_num; int B, L;    // taps [k][512], token_num [B]
  float* x;                                                           // residual stream [B L, 512] fp32, updated in place
  const float* n3_g; const float* n3_b;                               // norm3
  const half_t* Wqt; const float* bq; float qscale; half_t* q16; int ldq;   // launch_ffn_retile_out image of Wq; q out f16 [B L, ldq]
};
bool launch_dec_mid(hipStream_t s, const DecMidArgs& a);              // false: geometry not covered (caller runs the three launches)
size_t ffn_dec_image_bytes();
size_t ffn_dec_workspace_bytes(int M, int splits = 0);
int ffn_dec_splits(int M);
void launch_ffn_dec_retile(hipStream_t s, const half_t* W1, int ldw1, const float* W2_f32, const float* gamma, const float* beta,
                           const float* b1, half_t* img);
void launch_ffn_dec(hipStream_t s, const FfnDecArgs& a);

// ---------------------------------------------------------------- fp32 parity mode (k_fp32.hip) ----
void launch_gemm_f32(hipStream_t s, const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K,
                     float* out, int ldc, const float* resid, int ldr, bool relu, int scale_cols, float scale);
void launch_attention_f32(hipStream_t s, const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs,
                          const float* v, int64_t v_bs, int v_rs, float* o, int64_t o_bs, int o_rs, int B, int H, int Lq, int Lk);
bool launch_attention_f32_pair(hipStream_t s, const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs,
                               const float* v, int64_t v_bs, int v_rs, half_t* pair, int64_t p_bs, int p_rs, int p_lo, int B, int H, int Lq, int Lk);
// x [rows, K] fp32 -> out [rows, 2 Kp] f16 = hi | lo' (swap: lo' | hi), lo' = f16((x - hi) * 2^11); Kp % 4 == 0, pad columns zeroed
void launch_split_x3(hipStream_t s, const float* x, int64_t rows, int K, int ldx, half_t* out, int ldo, int Kp, int swap);
// the same attention with x3 operands (22-bit pairs on the f16 matrix cores): math_mode 3
void launch_attention_x3(hipStream_t s, const float* q, int64_t q_bs, int q_rs, const float* k, int64_t k_bs, int k_rs,
                         const float* v, int64_t v_bs, int v_rs, float* o, int64_t o_bs, int o_rs, int B, int H, int Lq, int Lk, bool scores_f32 = false);
void launch_im2col_f32(hipStream_t s, const float* H, int B, int T, int D, int l_order, int r_order, float* out);
void launch_add_f32(hipStream_t s, float* x, const float* y, int64_t n);     // x += y
void launch_spin(hipStream_t s, unsigned long long cycles);                  // one wave spinning ~cycles shader clocks (queue probe)
void launch_nop(hipStream_t s);
void launch_export_plan(hipStream_t s, const int32_t* max_count, const int32_t* fire_count, const int32_t* token_num, int B, int32_t* dst_dev);
void launch_posenc_f32(hipStream_t s, const float* x, const float* pe, int B, int T, int F, float xscale, float* out);
// one LSTM time step on fp32 gate pre-activations [B, 4D] (i, f, g, o): c / h [B, D] updated in place, h also to hout
void launch_lstm_cell_f32(hipStream_t s, const float* gates, int ldg, float* c, float* h, float* hout, int64_t hout_bs, int B, int D);

// ------------------------------------------------------------- frontend -----
struct FbankTables;   // device tables (window, twiddles, mel weights)
FbankTables* fbank_tables_create(int n_mels, int fs, const char* window);
void fbank_tables_destroy(FbankTables*);
// audio: B device pointers are expressed as base + offsets; out [sum T80, n_mels]
void launch_fbank(hipStream_t s, const FbankTables* tb, const float* audio, const int64_t* audio_off,
                  const int64_t* n_samples, const int64_t* frame_off, int B, int64_t total_frames,
                  int snip_edges, float* fbank, float dither = 0.f, uint32_t dither_seed = 0);
// LFR (m,n) + CMVN + right pad + sentinel: fbank rows (per-utt offsets) -> [B,Tmax,m*80]
void launch_lfr_cmvn_pad(hipStream_t s, const float* fbank, const int64_t* frame_off, const int32_t* t80,
                         int B, int Tmax, int lfr_m, int lfr_n, int n_mels, const float* shift,
                         const float* scale, int apply_cmvn, int apply_sentinel, float* out, const float* prompt = nullptr, int P = 0);
// ragged features -> padded + sentinel (PadSequence)
void launch_pad_sentinel(hipStream_t s, const float* feats, const int64_t* feat_off, const int32_t* n_floats,
                         int B, int row_floats, float* out);

// ----------------------------------------------------------------- norms ----
// x*sqrt(d_model) + PE(pos 1..T) then LayerNorm(width F) -> f16 [M, ldo] (cols F..ldo-1 zeroed)
void launch_posenc_ln_tab(hipStream_t s, const float* speech, int B, int T, int F, float xscale, const float* pe,
                      const float* gamma, const float* beta, half_t* out, int ldo);
// LayerNorm rows of width D (512 or 2048, or generic) : fp32 in -> f16 and/or fp32 out
void launch_layernorm_pair(hipStream_t s, const float* x, int64_t rows, const float* gamma, const float* beta, half_t* out, int ldo,
                           int lo_off);     // D = 512; result as the (hi | lo') operand pair of an x3 product only (k_norm.hip)
void launch_layernorm(hipStream_t s, const float* x, int64_t rows, int D, const float* gamma,
                      const float* beta, half_t* out16, int ld16, float* out32, int ld32);

// -------------------------------------------------------------- attention ---
struct AttnArgs {
  const half_t* q; int64_t q_bstride; int q_rstride;   // element strides (batch, row)
  const half_t* k; int64_t k_bstride; int k_rstride;
  const half_t* v; int64_t v_bstride; int v_rstride;
  half_t* o; int64_t o_bstride; int o_rstride;
  int B, H, Lq, Lk;                                   // head dim fixed at 128
  float* range;                                       // null, or quant_scratch_bytes(): {min, max} of the stored context per workgroup
                                                      // (pass 1 of the quantiser that consumes it, k_quant.hip); see attention_reports_range
  // Q and K handed over in ONE blocked matrix (the result of launch_gemm_qkvp): q == k == its base, `blk_groups` 8-column
  // groups per row (128), utterance b starts at row b * blk_brows, head h's Q columns are groups 16 h .., its K columns
  // groups blk_kgrp + 16 h ..; the q / k strides are ignored, V and the output stay row-major
  int qk_blocked, blk_groups, blk_brows, blk_kgrp;
};
void launch_attention(hipStream_t s, const AttnArgs& a);
bool attention_reports_range(const AttnArgs& a);      // the launch has at most 256 workgroups (one {min, max} pair each)

// ------------------------------------------------------------------ misc ----
void launch_f32_to_f16(hipStream_t s, const float* x, int64_t rows, int cols, int ldx, half_t* y, int ldy);
// encoder FSMN: v f16 [B*T, ldv] (cols 0..D-1 used), w [D,k] -> f fp32 [B*T, D]: dwconv + identity
void launch_fsmn_enc(hipStream_t s, const half_t* v, int ldv, const float* w, int B, int T, int D, int k, float* f);
// decoder FSMN: x[B*L,D] += (dwconv(tn*m) + tn*m)*m ; tn fp32 [B*L,D]; valid l < token_num[b]
void launch_fsmn_dec(hipStream_t s, const float* tn, const float* w, const int32_t* token_num, int B, int L,
                     int D, int k, float* x);
// LayerNorm of f16 rows -> f16 (may run in place); D % 8 == 0, D <= 2048
void launch_layernorm_f16(hipStream_t s, const half_t* x, int64_t rows, int D, const float* gamma, const float* beta, half_t* out);
// the same followed by LayerNorm(x) -> f16 (norm3): one launch; false = geometry not covered (D != 512 or k not 11/21)
bool launch_fsmn_dec_ln(hipStream_t s, const float* tn, const float* w, const int32_t* token_num, int B, int L, int D, int k,
                        float* x, const float* gamma, const float* beta, half_t* out16);
// streaming decoder FSMN (FunASR MultiHeadedAttentionSANMDecoder export with a cache): xc = cat(cache_in [B,D,K-1],
// (tn*m)^T); x[b,l,:] += (sum_j w_j * xc[l+j] + tn[l]*m) * m, m = (l < len[b]); cache_out = last K-1 columns of xc
void launch_fsmn_dec_stream(hipStream_t s, const float* tn, const float* wT, const int32_t* len, const float* cache_in,
                            int B, int L, int D, int k, float* x, float* cache_out);
// generic fp32 FSMN for the stand-alone op (mask [B,T] floats or null)
void launch_fsmn_f32_ld(hipStream_t s, const float* v, int ldv, const float* wT, int B, int T, int D, int k, float* y);
void launch_fsmn_f32(hipStream_t s, const float* v, const float* w, const float* mask, int B, int T, int D,
                     int k, float* y);
// CIF: im2col of H (f16) for the k=3 conv: [B*T, 3*D]
void launch_cif_im2col(hipStream_t s, const half_t* H, int B, int T, int D, int l_order, int r_order, half_t* out);
// alphas[b,t] = relu(sigmoid(dot(y[b,t,:], w) + b0)*smooth - noise), alphas[b,T] = tail
void launch_cif_alpha(hipStream_t s, const float* y, int B, int T, int D, const float* w, const float* b0,
                      float smooth, float noise, float tail, float* alphas);
// sequential integrate-and-fire per utterance -> fire table; then weighted gather
struct CifPlan {           // device arrays sized for B utterances
  int32_t* fire_count;     // [B]
  int32_t* token_num;      // [B]
  int32_t* fire_frame;     // [B, T1]  frame index of the l-th fire
  float* w_cur;            // [B, T1]  weight applied to frame t in the token open at t
  float* w_rem;            // [B, T1]  remainder carried into the next token when t fires
  int32_t* max_count;      // [1]
};
void launch_cif_scan(hipStream_t s, const float* alphas, int B, int T1, float threshold, CifPlan plan);
void launch_cif_gather(hipStream_t s, const float* H, int B, int T, int D, int T1, CifPlan plan, int L,
                       float* E);
// the prefix-sum CIF formulation (FunASR cif_v1_export): fire table + carried remainders, then the embeddings
void launch_cif_scan_cumsum(hipStream_t s, const float* alphas, int B, int T1, CifPlan plan);
void launch_cif_gather_cumsum(hipStream_t s, const float* H, const float* alphas, int B, int T, int D, int T1, CifPlan plan,
                              int L, float* E);
// ---------------------------------------------------- BiCIF timestamp head ----
struct LstmArgs {
  const half_t* whh;   // [2 dir][4D][D] f16, PyTorch gate order i,f,g,o
  const float* xg;     // [B*T3][2 dir * 4D] input-side gate pre-activations (+ both biases)
  half_t* hstate;      // [ndir][2 ping-pong][B][D]; the persistent launcher needs room for [ndir][4][B][D] (ring form) and initialises it
  float* cstate;       // [2 dir][B][D]
  float* hout;         // [B*T3][2D]  forward | reverse hidden states
  int B, T3, D, step;  // step s handles t = s (forward) and t = T3-1-s (reverse)
  int ndir;            // 2 = bidirectional (BiCIF head), 1 = forward only (SeACo hotword embedder)
};
void launch_lstm_step(hipStream_t s, const LstmArgs& a);
// all T3 steps in one launch (a.step ignored; hstate / cstate zeroed by the caller; cstate is not used: the cell
// state lives in registers).  sync_words: >= 64 device words (arrival counters + [63] = time-out flag, zeroed here).
// false = not applicable (more workgroups than CUs): use launch_lstm_step per step.
bool launch_lstm_persistent(hipStream_t s, const LstmArgs& a, unsigned* sync_words);
// the same recurrence with (hi, lo') pair operands (math_mode 3): whh = [ndir][4D][2D] pair rows, hstate [ndir][4][B][2D]
bool launch_lstm_persistent_x3(hipStream_t s, const LstmArgs& a, unsigned* sync_words);
void launch_us_alpha(hipStream_t s, const float* hout, int64_t rows, int W, const float* w, const float* b0,
                     float smooth, float noise, float* out);
void launch_us_peak(hipStream_t s, float* alphas, const int32_t* token_num, int B, int T3, float thr, float* peak);
// ------------------------------------------------------------------ SeACo ----
// rows[r, :] = table[ids[r], :]  (fp32) and its f16 copy
void launch_embed_gather(hipStream_t s, const float* table, const int32_t* ids, int rows, int D, int vocab,
                         float* out32, half_t* out16);
// out16[r, :] = f16(a[r, :] + b[r, :])
void launch_add_to_f16(hipStream_t s, const float* a, const float* b, int64_t rows, int D, half_t* out16);
// per row: keep = (first-index arg-max of dha[row, 0:V] == nobias); if !keep: ids[row] = dha_ids[row] and
// (copy_logits) logits[row, 0:V] = dha[row, 0:V]
void launch_seaco_merge(hipStream_t s, const float* dha, int ld_dha, const int64_t* dha_ids, int64_t rows, int V,
                        int nobias, int copy_logits, float* logits, int ld_logits, int64_t* ids);
// last-index arg-max over rows of width V.  mode 0: over the values as given; 1: over the log-probs
// y = (x - max) - log(sum exp(x - max)) (what the reference scans), y not stored; 2: same, y stored in place
void launch_argmax(hipStream_t s, float* x, int64_t rows, int V, int ldx, int mode, int64_t* ids);

}  // namespace pf
