/*
 * paraformer_hip.h — C ABI of libparaformer_hip.so
 *
 * MI355X (gfx950) native replacement for the two native engines behind the
 * reference's offline path (manyeyes/AliParaformerAsr; citations are relative
 * to that repository's root):
 *
 *   - Microsoft.ML.OnnxRuntime  InferenceSession.Run  (encoder + CIF + decoder)
 *       call sites AliParaformerAsr/OfflineProjOfParaformer.cs:68,
 *                  AliParaformerAsr/OfflineProjOfSenseVoiceSmall.cs:156,
 *                  AliParaformerAsr/OfflineProjOfSeacoParaformer.cs:116
 *   - ManySpeech.SpeechFeatures OnlineFbank.GetFbank  (kaldi fbank)
 *       call site  AliParaformerAsr/WavFrontend.cs:21-27,35
 *
 * plus the managed hot loops around them (LFR/CMVN WavFrontend.cs:53-111,
 * PadSequence Utils/PadHelper.cs:23-65, arg-max OfflineRecognizer.cs:139-152,
 * CIF-peak timestamps OfflineRecognizer.cs:200-302).
 *
 * Conventions
 *   - plain C types only; the caller allocates and pins every in/out buffer;
 *     the library never retains a caller pointer past return;
 *   - every function returns PF_OK (0) or a negative pf_status; the message is
 *     available from pf_last_error() (thread-local);
 *   - calls on one handle are serialised internally (one HIP stream per
 *     handle); different handles may be used from different threads;
 *   - there is no CPU fallback: without a usable gfx950 device pf_create
 *     fails with PF_ERR_DEVICE.
 *
 * The reference-side binding for each entry point (C# P/Invoke) is shown in
 * INTEGRATION.md.
 */
#ifndef PARAFORMER_HIP_H_
#define PARAFORMER_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 6

typedef enum pf_status {
  PF_OK = 0,
  PF_ERR_INVALID_ARG = -1,   /* null pointer, bad size; C# maps to ArgumentException          */
  PF_ERR_DEVICE = -2,        /* no gfx950 device / HIP runtime error                          */
  PF_ERR_IO = -3,            /* file missing or unreadable                                    */
  PF_ERR_FORMAT = -4,        /* malformed .pfw / am.mvn / yaml / tokens                       */
  PF_ERR_CAPACITY = -5,      /* caller buffer too small (required size is reported)           */
  PF_ERR_UNSUPPORTED = -6,   /* model kind / option not built                                 */
  PF_ERR_DISPOSED = -7,      /* handle used after dispose (ObjectDisposedException)           */
  PF_ERR_TOKENS = -8,        /* "tokens invalid" (OfflineRecognizer.cs:30-33)                 */
  PF_ERR_NULL_SAMPLES = -9,  /* ArgumentNullException("source") (WavFrontend.cs:34)           */
  PF_ERR_RECOGNITION = -10   /* "Offline recognition failed" (OfflineRecognizer.cs:194-197)   */
} pf_status;

int pf_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* pf_last_error(void);

/* ------------------------------------------------------------------------ */
/* 1. Engine handle: replaces OfflineModel (ORT session, OfflineModel.cs:35-70)
 *    + the per-stream WavFrontend state (WavFrontend.cs:18-29).              */
/* ------------------------------------------------------------------------ */
typedef struct pf_engine pf_engine;

typedef struct pf_engine_config {
  int32_t struct_size;        /* = sizeof(pf_engine_config)                                   */
  int32_t device;             /* HIP device ordinal                                           */
  /* weights: exactly one of the three sources (PFW1 container, see weights.py) */
  const char* weights_path;   /* file                                                          */
  const void* weights_host;   /* host image                                                    */
  const void* weights_device; /* device image (e.g. landed by an RCCL broadcast); not copied,
                                 must outlive the engine                                       */
  int64_t weights_bytes;      /* size of the host/device image                                 */
  /* CMVN: am.mvn path (WavFrontend.cs:112-153) or explicit vectors            */
  const char* mvn_path;
  const float* cmvn_shift;    /* <AddShift> vector                                             */
  const float* cmvn_scale;    /* <Rescale> vector                                              */
  int32_t cmvn_dim;           /* 560                                                           */
  /* FrontendConfEntity (Model/FrontendConfEntity.cs:7-15); 0 / NULL = reference default,
     except dither whose default is carried explicitly                         */
  int32_t fs;                 /* 16000                                                         */
  int32_t n_mels;             /* 80                                                            */
  int32_t lfr_m;              /* 7                                                             */
  int32_t lfr_n;              /* 6                                                             */
  int32_t snip_edges;         /* 0 (false)                                                     */
  float dither;               /* kaldi dither: N(0,1) * dither added to every sample of every frame
                                 (reference default 1.0, Model/FrontendConfEntity.cs:12); drawn on the
                                 device from a counter-based generator seeded by dither_seed          */
  const char* window;         /* "hamming"                                                     */
  int32_t use_itn;            /* SenseVoice: conf.use_itn (OfflineRecognizer.cs:27)            */
  int32_t frame_length_ms;    /* 0 = 25; anything but 25 -> PF_ERR_UNSUPPORTED                 */
  int32_t frame_shift_ms;     /* 0 = 10; anything but 10 -> PF_ERR_UNSUPPORTED                 */
  int32_t dither_seed;        /* seed of the dither stream (same seed + same audio = same features) */
  int32_t math_mode;          /* 0 = f16 operands on the MFMA (default); 1 = fp32 MFMA parity mode
                                 (v_mfma_f32_32x32x2_f32: exact fp32 products, ~1/16 of the speed); 2 = the arithmetic
                                 of the reference's default model.int8.onnx (Examples/Program.cs:98-101): every
                                 Linear as DynamicQuantizeLinear + MatMulInteger on v_mfma_i32_32x32x32_i8, weights
                                 quantised per output channel as onnxruntime's quantize_dynamic does; 3 = "exact" at
                                 matrix-core speed (ABI 5): the fp32 graph of mode 1 with every large Linear as three
                                 f16 MFMA products of (hi, 2^11 lo) operand pairs — 22 mantissa bits per operand, fp32
                                 accumulation — and fp32-MFMA flash attention */
  int32_t reserved[3];
} pf_engine_config;

int pf_engine_create(const pf_engine_config* cfg, pf_engine** out);
/* Idempotent teardown (Dispose(bool) pattern, OfflineRecognizer.cs:448-476: Dispose() and a later finaliser
   may both call it); waits for a call in flight on another thread.  Later calls on the handle -> PF_ERR_DISPOSED. */
void pf_engine_destroy(pf_engine* e);

/* Model facts the managed side needs. */
int pf_engine_info(pf_engine* e, int32_t* kind /*0 paraformer,1 sensevoice,2 seaco*/,
                   int32_t* vocab, int32_t* feat_dim, int32_t* has_timestamp_head);

/* ------------------------------------------------------------------------ */
/* 2. Seam "WavFrontend": GetFbank + LfrCmvn (WavFrontend.cs:31-51), called
 *    from OfflineStream.AddSamples (OfflineStream.cs:40-41).                 */
/* ------------------------------------------------------------------------ */
/* Number of LFR frames the front-end yields for n samples (floor(T80/lfr_n), quirk Q1). */
int pf_frontend_num_frames(pf_engine* e, int64_t n_samples, int32_t* t_lfr_out);
/* samples in [-1,1] -> feats [T, lfr_m*n_mels] (row-major, float32).
   feats_cap = capacity of feats_out in floats. samples == NULL -> PF_ERR_NULL_SAMPLES. */
int pf_frontend(pf_engine* e, const float* samples, int64_t n_samples,
                float* feats_out, int64_t feats_cap, int32_t* t_lfr_out);
/* kaldi fbank only: [T80, n_mels] (OnlineFbank.GetFbank, WavFrontend.cs:35). */
int pf_fbank(pf_engine* e, const float* samples, int64_t n_samples,
             float* fbank_out, int64_t fbank_cap, int32_t* t80_out);

/* ------------------------------------------------------------------------ */
/* 3. Seam "IOfflineProj.ModelProj" (+ arg-max): IOfflineProj.cs:38,
 *    OfflineProjOfParaformer.cs:39-87, OfflineRecognizer.cs:139-152.         */
/* ------------------------------------------------------------------------ */
typedef struct pf_batch_out {
  int32_t struct_size;
  /* capacities (in) */
  int32_t l_cap;              /* token slots per utterance in token_ids                        */
  int64_t logits_cap;         /* floats available in logits (0 = do not return logits)         */
  int64_t cif_peak_cap;       /* floats available in cif_peak (0 = skip)                       */
  /* outputs */
  int64_t* token_ids;         /* [B, l_cap] arg-max ids, first L valid per row; ties -> larger
                                 index (quirk Q4); all L positions kept (quirk Q5)             */
  int32_t* token_num;         /* [B] model_out_lens (floor(sum alpha)); unused by reference    */
  float* logits;              /* [B, L, V] log-probs (model_out), optional                     */
  float* cif_peak;            /* [B, 3*Tmax] us_cif_peak, optional (timestamp models)          */
  int32_t L;                  /* out: decoder positions per utterance (logits dim 1)           */
  int32_t V;                  /* out: vocabulary (logits dim 2)                                */
  int32_t cif_peak_len;       /* out: 3*Tmax or 0                                              */
  int32_t reserved;
} pf_batch_out;

/* `speech` is the tensor the reference hands to ORT: [B, Tmax, feat_dim] float32,
   already padded and sentinel-substituted (PadHelper.cs:63); speech_lengths is
   Tmax for every row (quirk Q2) and therefore not a parameter.
   hotwords: SeACo only, int32 [n_hotwords, 10] (EmbedSeacoModel.cs:70-123), may be NULL. */
int pf_forward_feats(pf_engine* e, const float* speech, int32_t B, int32_t Tmax,
                     const int32_t* hotwords, int32_t n_hotwords, pf_batch_out* out);

/* ModelProj including PadSequence: B ragged feature buffers (OfflineInputEntity.Speech,
   SpeechLength = float count each) -> padded on device, sentinel applied, forward. */
int pf_model_proj(pf_engine* e, const float* const* speech, const int32_t* speech_len_floats,
                  int32_t B, const int32_t* hotwords, int32_t n_hotwords, pf_batch_out* out);

/* ------------------------------------------------------------------------ */
/* 4. Fused fast path: raw audio in, ids out (AddSamples + GetResults numeric
 *    part in one device pipeline).                                           */
/* ------------------------------------------------------------------------ */
int pf_recognize(pf_engine* e, const float* const* samples, const int64_t* n_samples,
                 int32_t B, const int32_t* hotwords, int32_t n_hotwords, pf_batch_out* out);

/* Split form used by the benchmark so that the timed region starts with the audio
   resident in HBM: stage (H2D) -> run (device only, async on the engine stream) ->
   fetch (D2H of ids). */
int pf_stage_audio(pf_engine* e, const float* const* samples, const int64_t* n_samples, int32_t B);
/* SeACo: hotword ids [n_hotwords, 10] (PadList output, EmbedSeacoModel.cs:70-123) used by the following
   pf_run_staged calls (the other forward entry points take them per call); ignored by other model kinds. */
int pf_engine_set_hotwords(pf_engine* e, const int32_t* hotwords, int32_t n_hotwords);
int pf_run_staged(pf_engine* e);      /* enqueues the whole pipeline, returns after the CIF
                                         length read-back (the path's only host sync)          */
int pf_sync(pf_engine* e);            /* waits for the engine stream                           */
/* Copies the results of the calling THREAD's last pf_forward_feats / pf_model_proj / pf_recognize (kept per
   thread, so the two-call protocol — first call learns L and V, pf_fetch fills right-sized buffers — is safe
   with concurrent callers on one engine; calls themselves are serialised by an internal mutex, like the
   reference's static lock, OfflineStream.cs:19).  After pf_run_staged it returns the staged result (the
   staged API is engine state: one caller at a time). */
int pf_fetch(pf_engine* e, pf_batch_out* out);
/* The hypotheses of the staged result WITHOUT leaving the device (ABI 4): writes ids [B, l_cap] int64 (columns >= L
   filled with -1) into `ids_dev` (device memory of the engine's GPU) on the engine stream and waits for it, so a caller
   that gathers hypotheses across GPUs (RCCL all-gather over xGMI) hands the buffer straight to the
   collective.  *L_out (optional) receives the decoder length.  PF_ERR_CAPACITY when l_cap < L.  Replaces, for that
   caller, the host copy the reference makes of the logits tensor (OfflineProjOfParaformer.cs:73-79). */
int pf_fetch_ids_device(pf_engine* e, int64_t* ids_dev, int32_t l_cap, int32_t* L_out);

/* ------------------------------------------------------------------------ */
/* 4b. Multi-GPU inside one process: one engine, one host thread and one HIP stream per listed
 *     device.  The reference builds a single ORT session (OfflineRecognizer.cs:23); what shards is the utterance
 *     list handed to GetResults (OfflineRecognizer.cs:110-116) — utterances are independent (the batch is dim 0 of
 *     every tensor, OfflineProjOfParaformer.cs:49).  RCCL (dlopen'ed librccl) carries the one-off weight broadcast
 *     devices[0] -> all and the per-call all-gather of the fixed-shape hypotheses; there is no data-path collective.
 *     Every shard is padded to the batch-wide maximum length (PadHelper.cs:25) and decodes the batch-wide maximum
 *     token count, so ids / token_num / cif_peak equal the single-device result position by position.
 *     A device may be listed more than once (several engines on one GPU; no communicator is created then).     */
/* ------------------------------------------------------------------------ */
typedef struct pf_group pf_group;
int pf_group_create(const pf_engine_config* cfg /* .device ignored */, const int32_t* devices, int32_t n_devices,
                    pf_group** out);
void pf_group_destroy(pf_group* g);               /* idempotent, like pf_engine_destroy                       */
int pf_group_info(pf_group* g, int32_t* n_engines, int32_t* uses_rccl);
/* Engine i of the group (borrowed: valid until pf_group_destroy), e.g. for pf_engine_info / profiling.      */
pf_engine* pf_group_engine(pf_group* g, int32_t i);
/* pf_recognize over the whole group: contiguous shards of ceil(B / n_engines) utterances run concurrently, the
   result comes back in the caller's order.  Same two-call protocol as pf_recognize + pf_group_fetch.          */
int pf_group_recognize(pf_group* g, const float* const* samples, const int64_t* n_samples, int32_t B,
                       const int32_t* hotwords, int32_t n_hotwords, pf_batch_out* out);
int pf_group_fetch(pf_group* g, pf_batch_out* out);

/* Host-only rehearsal of pf_group_recognize's control flow (shard plan, the three rendez-vous, the fixed-shape
   all-gather blocks, failure release, merge in the caller's order) with arithmetic stand-ins for the devices, so
   that it runs in a CPU test-suite with G up to 64: utterance u "fires" fire_count[u] tokens and decodes to
   ids[u][l] = u * 100000 + l, token_num[u] = fire_count[u].  has_cif = 0: every non-empty shard reports L = fixed_L
   on its own (SenseVoice).  collective != 0: the hypotheses travel through the all-gather blocks, and shards entering
   the collective with different block sizes — which RCCL answers with a hang — are an error.  fail_shard >= 0: that
   shard throws at fail_stage (0 = in its forward before the decoder-length rendez-vous, 1 = after it, 2 = while
   preparing the gather); the call must then return an error, never hang. */
int pf_host_group_sim(int32_t G, int32_t B, const int32_t* fire_count, int32_t has_cif, int32_t fixed_L,
                      int32_t collective, int32_t fail_shard, int32_t fail_stage, int64_t* ids_out, int32_t l_cap,
                      int32_t* token_num_out, int32_t* L_out);

/* Per-kernel-class device time, measured with HIP events on the engine stream while
   profiling is enabled (bench.py roofline leg).  class_name e.g. "gemm_ffn1". */
int pf_profile_enable(pf_engine* e, int32_t on);
int pf_profile_reset(pf_engine* e);
/* Restrict event recording to one class (NULL or "" = all classes). */
int pf_profile_select(pf_engine* e, const char* class_name);
int pf_profile_get(pf_engine* e, const char* class_name, double* total_ms, int64_t* launches,
                   double* flops_per_launch);
/* GEMM classes: name of the kernel the launcher chose for the class's last profiled launch (the name the
   rocprofv3 kernel trace shows), "" for other classes.  cap = bytes available in name_out. */
int pf_profile_kernel(pf_engine* e, const char* class_name, char* name_out, int32_t cap);

/* Algorithmic FLOPs (2*MAC) of the last forward, SURVEY.md §8(d) formula. */
int pf_last_flops(pf_engine* e, double* flops);

/* ------------------------------------------------------------------------ */
/* 5. Stand-alone device ops exposed for parity tests (tests/ call these through
 *    the C ABI).  pf_op_gemm_ex / pf_op_gemm_rc / pf_op_ffn / pf_op_fsmn_enc / pf_op_fsmn_dec /
 *    pf_op_logsoftmax_argmax / pf_op_attention / pf_op_layernorm / pf_op_cif /
 *    pf_op_lfr_cmvn_pad launch exactly the kernels (and kernel variants) the
 *    pipeline launches; pf_op_gemm chooses its variant by shape like the
 *    pipeline does; pf_op_fsmn is a generic fp32 FSMN (arbitrary mask) that the
 *    pipeline itself does not launch; pf_op_argmax scans the values as given. */
/* ------------------------------------------------------------------------ */
/* LFR + CMVN + right-pad + sentinel: fbank rows of B utterances -> [B, Tmax, lfr_m*80]. */
int pf_op_lfr_cmvn_pad(pf_engine* e, const float* const* fbank, const int32_t* t80, int32_t B,
                       int32_t apply_sentinel, float* out, int64_t out_cap, int32_t* tmax_out);
/* last-index arg-max over the trailing dim: x [rows, V] -> ids [rows]. */
int pf_op_argmax(pf_engine* e, const float* x, int64_t rows, int32_t V, int64_t* ids_out);
/* C = A[M,K] * W[N,K]^T + bias, f16 operands / f32 accumulate; epilogue 0 none, 1 relu,
   2 = f16 result store (the path the pipeline uses), returned widened to fp32. */
/* One dynamically quantised Linear, the building block of math_mode 2 (the reference's default model.int8.onnx:
   DynamicQuantizeLinear + MatMulInteger + rescale, as onnxruntime's quantize_dynamic emits them for every MatMul with a
   constant weight).  x [M, K] fp32 (x_is_f16: first rounded to f16, as the engine's f16-stored activations are),
   W [N, K] fp32 (quantised per output channel to uint8 here), y [M, N] = float(sum (x_q - x_zp)(w_q - w_zp[n])) *
   (x_scale * w_scale[n]) + bias[n] [ReLU]; the integer sum runs on v_mfma_i32_32x32x32_i8 and is exact.  Optional outputs
   (NULL = skip): the uint8 activations [M, K], {x_scale, x_zp}, the uint8 weights [N, K], w_scale [N], w_zp [N].
   x_is_f16 is a bit set: 1 = round x to f16 first; 2 = run the f16-result kernel of the pipeline's QKV / FFN-up
   projections (y = the stored f16 values widened; PF_ERR_DEVICE if the range its epilogue reports for the next
   quantiser differs from a min / max pass over its output). */
int pf_op_qlinear(pf_engine* e, const float* x, const float* W, const float* bias, int32_t M, int32_t N, int32_t K,
                  int32_t relu, int32_t x_is_f16, float* y, uint8_t* xq_out, float* aparams_out, uint8_t* wq_out,
                  float* wscale_out, int32_t* wzp_out);
int pf_op_gemm(pf_engine* e, const float* A, const float* W, const float* bias,
               int32_t M, int32_t N, int32_t K, int32_t epilogue, float* C);
/* The GEMM as the pipeline launches it, every variant selectable. */
typedef struct pf_gemm_desc {
  int32_t struct_size;
  int32_t M, N, K;
  int32_t relu;
  int32_t out_kind;           /* 0 fp32 result (+ residual / addend), 1 f16 row-major, 2 f16 blocked layout
                                 (32 rows x 8 columns = 512 contiguous bytes; the encoder FFN hidden)         */
  int32_t a_blocked;          /* A is handed to the kernel in the blocked layout (FFN-down's operand)         */
  int32_t tile_rows;          /* kernel selector: 0 = by shape as the pipeline does (M <= 512 rows: the short-input
                                 kernel; blocked results with fewer idle rounds: the persistent 256 x 256 kernel);
                                 128 / 256 = gemm_f16_pp3 tile heights; 512 = the 256 x {192,256} tile kernel, one tile
                                 per workgroup; 1024 = its persistent form (blocked result only); 32 = the short-input
                                 kernel; 2048 = the k-step-32 six-stage kernel (fp32 results of deep-K projections, the
                                 encoder's FFN-down).  PF_ERR_INVALID_ARG when the named kernel does not apply.     */
  int32_t scale_cols;         /* columns n < scale_cols are multiplied by scale after the bias (q scaling)    */
  float scale;
  const float* bias;          /* [N] or NULL                                                                  */
  const float* resid;         /* [M,N] fp32 or NULL (out_kind 0)                                              */
  const float* add2;          /* [M,N] fp32 or NULL (out_kind 0): the FSMN memory added by the out-projection */
} pf_gemm_desc;
/* C [M,N] fp32 (f16 results widened, blocked results de-blocked on the host). */
int pf_op_gemm_ex(pf_engine* e, const pf_gemm_desc* d, const float* A, const float* W, float* C);
/* Row-complete GEMM (N = 512) with its fused epilogue, as the encoder launches it for the attention output
   projection and the FFN down-projection: x = resid + A W^T + bias + FSMN_k(v) ; n = LayerNorm(x). */
typedef struct pf_gemm_rc_desc {
  int32_t struct_size;
  int32_t M, K;
  int32_t a_blocked;          /* A handed over in the blocked activation layout                                */
  int32_t T;                  /* utterance length in rows (FSMN zero padding at utterance edges); 0 = M        */
  int32_t fsmn_k;             /* taps of fsmn_w (11), 0 = no FSMN term                                          */
  const float* bias;          /* [512] or NULL                                                                 */
  const float* resid;         /* [M,512] or NULL                                                               */
  const float* fsmn_v;        /* [M,512] (rounded to f16 on the device, as the V slice is) or NULL             */
  const float* fsmn_w;        /* [512, fsmn_k]                                                                 */
  const float* ln_gamma;      /* [512] or NULL (no LayerNorm outputs)                                          */
  const float* ln_beta;
  int32_t short_input;        /* 1 = the kernels the pipeline uses for M <= 512 rows (k_gemm_small.hip): K <= 576:
                                 one-shot GEMM with the FSMN memory as an epilogue term, then the LayerNorm kernel;
                                 K > 576: split partials + the reduction that carries the LayerNorm (no FSMN term) */
  int32_t split_k;            /* must be 0: the split-K forms of round 4 (k_gemm_sk.hip) lost to the fused FFN block and
                                 were removed in round 5 (PF_ERR_UNSUPPORTED); the field keeps the struct layout     */
} pf_gemm_rc_desc;
/* x_out [M,512] fp32 (may be NULL), n16_out [M,512] (the f16 LayerNorm result widened to fp32, may be NULL),
   n32_out [M,512] fp32 LayerNorm result (may be NULL). */
int pf_op_gemm_rc(pf_engine* e, const pf_gemm_rc_desc* d, const float* A, const float* W, float* x_out,
                  float* n16_out, float* n32_out);
/* Encoder FFN with the blocked hand-off of the hidden: y = resid + W2 relu(W1 x + b1) + b2;
   x [M,D], w1 [F,D], w2 [D,F], resid / y [M,D]. */
int pf_op_ffn(pf_engine* e, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
              const float* resid, int32_t M, int32_t D, int32_t F, float* y);
/* The same block as the pipeline launches it for long inputs since round 5 (k_ffn.hip, ONE launch, the hidden stays in
   LDS; d_model 512, hidden 2048): x_out = resid + W2 relu(W1 x + b1) + b2 [M,512] (may be NULL), n16_out = the f16
   LayerNorm(x_out; gamma, beta) widened to fp32 (NULL, or with gamma / beta).  resid may be NULL (zeros). */
int pf_op_ffn_fused(pf_engine* e, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* resid, const float* ln_gamma, const float* ln_beta, int32_t M, float* x_out, float* n16_out);
/* The decoder's position-wise block as the pipeline runs it since round 5 (k_ffn.hip, split form; d_model 512, hidden 2048):
   t = LayerNorm_F(relu(x w1^T + b1); gamma_f, beta_f) w2^T  (the w_1 / norm / w_2 nodes of a decoder layer, w2 without bias),
   n = LayerNorm(t; ln_gamma, ln_beta).  x [M,512] = the block's normalised input (rounded to f16 like every f16-mode operand) —
   or, with ctx != NULL, the PREVIOUS layer's cross-attention out-projection runs in front of the block in the same launch:
   x_out = resid + ctx wo^T + bo, the block's input = LayerNorm(x_out; ln1_gamma, ln1_beta) (x is then ignored).
   splits: how many workgroups share a 64-row tile's hidden range, 0 = the pipeline's choice for M, or 1 | 2 | 3 | 4 | 8. */
typedef struct pf_dec_ffn_desc {
  int32_t struct_size;           /* sizeof(pf_dec_ffn_desc) */
  int32_t M, splits, reserved;   /* reserved = 0 */
  const float* x;                /* [M,512] or NULL with ctx */
  const float* w1; const float* b1;            /* [2048,512], [2048] */
  const float* gamma_f; const float* beta_f;   /* [2048] */
  const float* w2;                             /* [512,2048] */
  const float* ln_gamma; const float* ln_beta; /* [512] or NULL */
  const float* ctx; const float* wo; const float* bo; const float* resid;   /* [M,512], [512,512], [512], [M,512] or all NULL */
  const float* ln1_gamma; const float* ln1_beta;                           /* [512], with ctx */
} pf_dec_ffn_desc;
/* t_out / n_out / x_out [M,512] fp32, each may be NULL (n_out needs ln_gamma / ln_beta, x_out needs ctx). */
int pf_op_dec_ffn_fused(pf_engine* e, const pf_dec_ffn_desc* d, float* t_out, float* n_out, float* x_out);
/* The same launch with the attention out-projection in front of the block, as the pipeline runs two thirds of an encoder
   layer since round 5: x_mid = resid + ctx wo^T + bo + FSMN(v) (11 taps, utterances = runs of T rows);
   x_out = x_mid + W2 relu(W1 LayerNorm(x_mid; ln2) + b1) + b2; n16_out = f16 LayerNorm(x_out; ln_gamma, ln_beta).
   ctx, v [M,512]; wo [512,512]; fsmn_w [512,11]; resid may be NULL. */
typedef struct pf_attn_ffn_desc {
  int32_t struct_size; int32_t M; int32_t T; int32_t reserved;
  const float* ctx; const float* wo; const float* bo; const float* v; const float* fsmn_w;
  const float* ln2_gamma; const float* ln2_beta; const float* resid;
  const float* w1; const float* b1; const float* w2; const float* b2;
  const float* ln_gamma; const float* ln_beta;
  /* optional tail, as the pipeline runs it between two encoder layers: the NEXT layer's fused Q | K | V projection of
     LayerNorm(x_out; ln_gamma, ln_beta) in the same launch: wqkv [1536,512] = [Q | K | V] rows, bqkv [1536]; q_out (scaled by
     1/sqrt(128)), k_out, v_out [M,512] = the stored f16 values widened to fp32 (each may be NULL) */
  const float* wqkv; const float* bqkv; float* q_out; float* k_out; float* v_out;
} pf_attn_ffn_desc;
int pf_op_attn_ffn_fused(pf_engine* e, const pf_attn_ffn_desc* d, float* x_out, float* n16_out);
/* A Linear of the fp32 graph as the engine's math_mode runs it (engines created with math_mode 1 or 3 only):
     y [M,N] = x [M,K] W[N,K]^T + bias [+ resid [M,N]] [ReLU]
   math_mode 3 ("exact"): operands as (hi, lo') f16 pairs, the three partial products in ONE accumulation of the pipeline's
   f16 MFMA kernel (K-loop wrap, csrc/kernels.h) — the stand-alone counterpart of Engine::gemm32.  bias / resid may be NULL. */
int pf_op_linear32(pf_engine* e, const float* x, const float* W, const float* bias, const float* resid, int32_t M, int32_t N,
                   int32_t K, int32_t relu, float* y);
/* The FFN block of the fp32 graph as math_mode 1 / 3 runs it: y [M,D] = x + relu(x W1^T + b1) W2^T + b2, W1 [F,D], W2 [D,F].
   In math_mode 3 above the short-input threshold the hidden never exists in fp32: the first product's epilogue writes it as
   the (hi, lo') operand pair of the second (what the encoder layers do). */
int pf_op_ffn32(pf_engine* e, const float* x, const float* W1, const float* b1, const float* W2, const float* b2, int32_t M,
                int32_t D, int32_t F, float* y);
/* Encoder FSMN kernel (f16 V slice of a [B*T, 3D] buffer in, fp32 out): y = dwconv_k(v) + v. */
int pf_op_fsmn_enc(pf_engine* e, const float* v, const float* w, int32_t B, int32_t T, int32_t D, int32_t k, float* y);
/* Decoder FSMN kernel: x += (dwconv_k(tn*m) + tn*m)*m, m = (l < token_num[b]); x in/out [B,L,D]. */
int pf_op_fsmn_dec(pf_engine* e, const float* tn, const float* w, const int32_t* token_num, int32_t B, int32_t L,
                   int32_t D, int32_t k, float* x);
/* The pipeline's vocabulary tail: y = log_softmax(x) (two-step form, see k_misc.hip) and the last-index arg-max
   over y — what OfflineRecognizer.cs:139-152 scans.  y_out may be NULL (ids-only variant of the kernel). */
int pf_op_logsoftmax_argmax(pf_engine* e, const float* x, int64_t rows, int32_t V, float* y_out, int64_t* ids_out);
/* LayerNorm over the last dim (eps 1e-12), fp32. */
int pf_op_layernorm(pf_engine* e, const float* x, const float* gamma, const float* beta,
                    int64_t rows, int32_t dim, float* y);
/* softmax(q k^T) v per head, q pre-scaled; q [B,Lq,H*128], k,v [B,Lk,H*128] (fp32 host,
   converted to f16 on device as the pipeline does). */
int pf_op_attention(pf_engine* e, const float* q, const float* k, const float* v,
                    int32_t B, int32_t Lq, int32_t Lk, int32_t heads, float* out);
/* The encoder's fused Q | K | V projection (d_model 512, 4 heads) and its self-attention as the pipeline launches them
   for long inputs: persistent 256 x 192 GEMM (q scaled by 1/sqrt(128); Q and K in the blocked activation layout, V
   row-major) + the attention kernel reading that layout.  x [B*T, K], w [1536, K] = [Q | K | V] rows, bias [1536] or
   NULL; q/k/v/ctx_out [B*T, 512] (the stored f16 values widened to fp32; each may be NULL). */
int pf_op_qkv_attention(pf_engine* e, const float* x, const float* w, const float* bias, int32_t B, int32_t T, int32_t K,
                        float* q_out, float* k_out, float* v_out, float* ctx_out);
/* DFSMN memory block: y = dwconv_k(v*mask) + v*mask, *mask; v [B,T,D], w [D,k], mask [B,T] or NULL. */
int pf_op_fsmn(pf_engine* e, const float* v, const float* w, const float* mask,
               int32_t B, int32_t T, int32_t D, int32_t k, float* y);
/* CIF integrate-and-fire: H [B,T,D], alphas [B,T+1] -> embeds [B,Lcap,D] (zero padded),
   fire_count [B], token_num [B]; returns L = max fire_count in *L_out. */
int pf_op_cif(pf_engine* e, const float* H, const float* alphas, int32_t B, int32_t T, int32_t D,
              float threshold, int32_t Lcap, float* embeds, int32_t* fire_count,
              int32_t* token_num, int32_t* L_out);
/* Encoder only: speech [B,T,feat] -> H [B,T,512] fp32. */
int pf_op_encoder(pf_engine* e, const float* speech, int32_t B, int32_t T, float* H);

/* ------------------------------------------------------------------------ */
/* 6. Host-side recognizer: same surface as the reference's public classes
 *    OfflineRecognizer (OfflineRecognizer.cs:13-477) and OfflineStream
 *    (OfflineStream.cs:7-121).  Implemented in C++ above the engine.         */
/* ------------------------------------------------------------------------ */
typedef struct pf_recognizer pf_recognizer;
typedef struct pf_stream pf_stream;

/* new OfflineRecognizer(modelFilePath, configFilePath, mvnFilePath, tokensFilePath,
                         modelebFilePath = "", hotwordFilePath = "", batchSize = 1, threadsNum = 1)
   (OfflineRecognizer.cs:23).  modelFilePath names a .pfw container; batchSize/threadsNum are
   accepted and unused exactly as in the reference (quirk Q14); device is the extra argument. */
int pf_recognizer_create(const char* model_path, const char* config_path, const char* mvn_path,
                         const char* tokens_path, const char* modeleb_path,
                         const char* hotword_path, int32_t batch_size, int32_t threads_num,
                         int32_t device, pf_recognizer** out);
void pf_recognizer_dispose(pf_recognizer* r);   /* Dispose(): frees the engine; later calls -> PF_ERR_DISPOSED   */
void pf_recognizer_free(pf_recognizer* r);      /* Dispose() + drops the handle's reference; streams created from it
                                                   stay valid handles (their calls answer PF_ERR_DISPOSED)        */
pf_engine* pf_recognizer_engine(pf_recognizer* r);
/* The recognizer owns a POOL of engines on its device (round 5): GetResults / AddSamples calls of different threads run on
   different engines instead of queueing — the reference's GetResults is unlocked (OfflineRecognizer.cs:110-198).  Engines
   beyond the first are created when a call finds all of them busy, up to $PF_RECOGNIZER_ENGINES (default 2, 1..8); they share
   the device weight image.  Returns how many exist (>= 1), or PF_ERR_DISPOSED.  pf_recognizer_engine hands out engine 0. */
int pf_recognizer_num_engines(pf_recognizer* r);

int pf_recognizer_create_stream(pf_recognizer* r, pf_stream** out);     /* CreateOfflineStream :92 */
/* AddSamples, OfflineStream.cs:36.  `samples` belongs to the caller again when the call returns (the reference's contract): an
   array the recognizer has not seen before has been copied into pinned staging memory by then and its DMA may still be in
   flight — whatever reads the samples later (pf_recognizer_get_results, a second pf_stream_add_samples, pf_stream_free) waits
   for it by itself. */
int pf_stream_add_samples(pf_stream* s, const float* samples, int64_t n);
/* stream.Hotwords = List<int[]> (flattened ids + per-hotword lengths); n_hotwords < 0 sets null. */
int pf_stream_set_hotwords(pf_stream* s, const int32_t* ids, const int32_t* lens, int32_t n_hotwords);
int pf_stream_get_hotwords(pf_stream* s, int32_t* ids, int32_t ids_cap, int32_t* lens,
                           int32_t lens_cap, int32_t* n_hotwords /* -1 = null */);
int pf_stream_num_feature_floats(pf_stream* s, int32_t* n);  /* OfflineInputEntity.SpeechLength */
void pf_stream_dispose(pf_stream* s);            /* DisposeOfflineStream :441: drops the buffers; the handle stays
                                                    valid and later calls answer PF_ERR_DISPOSED           */
void pf_stream_free(pf_stream* s);               /* releases the handle (and its share of the recognizer).  Idempotent: the
                                                    handle keeps answering PF_ERR_DISPOSED (Dispose + finaliser both
                                                    calling it is safe) until 65 536 younger stream handles have been
                                                    freed, after which its few dozen bytes are handed out again — a
                                                    server creating one stream per utterance does not grow          */

/* GetResults(List<OfflineStream>) (OfflineRecognizer.cs:110): Forward + DecodeMulti.
   Results stay owned by the recognizer until the next GetResults call / dispose. */
int pf_recognizer_get_results(pf_recognizer* r, pf_stream* const* streams, int32_t n_streams);
/* Accessors for result i of the last GetResults: OfflineRecognizerResultEntity
   {Text, TextLen, Tokens, Timestamps} (Model/OfflineRecognizerResultEntity.cs:9-29). */
int pf_result_text(pf_recognizer* r, int32_t i, const char** utf8, int32_t* text_len_utf16);
int pf_result_num_tokens(pf_recognizer* r, int32_t i, int32_t* n);
int pf_result_token(pf_recognizer* r, int32_t i, int32_t j, const char** utf8);
/* timestamp j of result i: n_ints is 2, or 4+ for merged BPE tokens (quirk Q10). */
int pf_result_timestamp(pf_recognizer* r, int32_t i, int32_t j, const int32_t** ints, int32_t* n_ints);
int pf_result_num_timestamps(pf_recognizer* r, int32_t i, int32_t* n);
/* stream.Tokens after Forward (raw ids, OfflineRecognizer.cs:187). */
int pf_stream_tokens(pf_stream* s, const int64_t** ids, int32_t* n);

/* ABI 6: the rest of OfflineStream's public surface (OfflineStream.cs:20-34).  Only the reference's own Forward touches
   these members, but "same public signatures" means a caller may.
   new OfflineStream(mvnFilePath, confEntity) (:20-28): a stream that belongs to no recognizer yet; the arguments are
   confEntity.frontend_conf's fields (0 / NULL = the FrontendConfEntity default).  Its AddSamples calls are kept on the host
   and replayed by the first GetResults that receives it; a front-end other than that recognizer's -> PF_ERR_UNSUPPORTED
   (raised by that GetResults inside Forward's try block: "Offline recognition failed"). */
int pf_stream_create(const char* mvn_path, int32_t fs, int32_t n_mels, int32_t lfr_m, int32_t lfr_n, int32_t snip_edges,
                     float dither, const char* window, pf_stream** out);
int pf_stream_set_tokens(pf_stream* s, const int64_t* ids, int32_t n);          /* Tokens { set; }      :32 */
int pf_stream_num_timestamps(pf_stream* s, int32_t* n);                          /* Timestamps { get; }  :33 */
int pf_stream_timestamp(pf_stream* s, int32_t j, const int32_t** ints, int32_t* n_ints);
int pf_stream_set_timestamps(pf_stream* s, const int32_t* ints, const int32_t* lens, int32_t n);   /* Timestamps { set; } */
/* OfflineInputEntity { Speech, SpeechLength } (:30; Model/OfflineInputEntity.cs).  get: *n_floats = -1 when Speech is null,
   else the float count (PF_ERR_CAPACITY with the count reported when cap is smaller); a stream whose samples live on the
   device is brought to the host form (features computed and read back).  set: n_floats < 0 sets Speech = null;
   speech_length is stored as given (the reference's two setters are independent). */
int pf_stream_get_speech(pf_stream* s, float* out, int64_t cap, int32_t* n_floats);
int pf_stream_set_speech(pf_stream* s, const float* speech, int32_t n_floats, int32_t speech_length);

/* ------------------------------------------------------------------------ */
/* 7. Streaming path: the reference's OnlineRecognizer / OnlineStream
 *    (AliParaformerAsr/OnlineRecognizer.cs:14-542, OnlineStream.cs:7-358).  The chunking, the feature caches,
 *    DynamicMask and the carried-integrator CIF are host code as in the reference (C++ here); the two ONNX
 *    sessions (OnlineModel.cs:23-31) are the device seams pf_online_encoder / pf_online_decoder.          */
/* ------------------------------------------------------------------------ */
typedef struct pf_online_recognizer pf_online_recognizer;
typedef struct pf_online_stream pf_online_stream;
/* new OnlineRecognizer(encoderFilePath, decoderFilePath, configFilePath, mvnFilePath, tokensFilePath, threadsNum)
   (OnlineRecognizer.cs:22): encoder_path names ONE .pfw container holding both graphs' tensors; decoder_path is
   accepted and unused; device is the extra argument. */
int pf_online_recognizer_create(const char* encoder_path, const char* decoder_path, const char* config_path,
                                const char* mvn_path, const char* tokens_path, int32_t threads_num, int32_t device,
                                pf_online_recognizer** out);
void pf_online_recognizer_dispose(pf_online_recognizer* r);
void pf_online_recognizer_free(pf_online_recognizer* r);
pf_engine* pf_online_recognizer_engine(pf_online_recognizer* r);
int pf_online_create_stream(pf_online_recognizer* r, pf_online_stream** out);          /* CreateOnlineStream :27 */
int pf_online_stream_add_samples(pf_online_stream* s, const float* samples, int64_t n); /* OnlineStream.AddSamples */
/* GetResults(List<OnlineStream>) (:41-48): one chunk per stream that has 60 fbank frames ready; texts are kept for
   the calling thread until its next call. */
int pf_online_get_results(pf_online_recognizer* r, pf_online_stream* const* streams, int32_t n_streams);
int pf_online_result_text(pf_online_recognizer* r, int32_t i, const char** utf8);
int pf_online_stream_tokens(pf_online_stream* s, const int64_t** ids, int32_t* n);
void pf_online_stream_dispose(pf_online_stream* s);
void pf_online_stream_free(pf_online_stream* s);
/* Device seams = the two InferenceSession.Run calls (OnlineRecognizer.cs:83, :296).
   encoder: speech [B,Tc,560] (scaled + position-encoded by the caller) -> enc [B,Tc,512], alphas [B,Tc].
   decoder: enc, acoustic_embeds [B,L,512] + lengths, in_cache [n_caches][B,512,10] -> log-probs [B,L,V] (optional),
            last-index arg-max ids [B,L], out_cache [n_caches][B,512,10]. */
int pf_online_encoder(pf_engine* e, const float* speech, int32_t B, int32_t Tc, float* enc_out, float* alphas_out);
int pf_online_decoder(pf_engine* e, const float* enc, int32_t B, int32_t Tc, const float* embeds, int32_t L,
                      const int32_t* embeds_len, const float* caches_in, int32_t n_caches, float* logits_out,
                      int64_t* ids_out, float* caches_out);
/* Host pieces exposed for parity tests (pure CPU): OnlineWavFrontend.ApplyLfr (:63-80), SinusoidalPositionEncoder
   (:152-188, in place), the per-stream CIF of PredictorProj (OnlineRecognizer.cs:152-197), DynamicMask
   (OnlineModel.cs:141-165, in place), DecodeMulti (:405-437). */
int pf_host_online_lfr(const float* fbank, int32_t t80, int32_t lfr_m, int32_t lfr_n, float* out, int64_t cap, int32_t* t_lfr);
int pf_host_online_posenc(float* x, int32_t timesteps, int32_t dim, int32_t start_idx);
int pf_host_online_dynamic_mask(float* alphas, int32_t n);
int pf_host_online_cif(const float* hiddens, const float* alphas, int32_t n, int32_t D, float threshold, float* fired,
                       int32_t fired_cap, int32_t* n_fired, float* carry_alpha, float* carry_hidden);
int pf_host_online_decode(const char* const* tokens, int32_t n_tokens, const int64_t* ids, int32_t n_ids, char* out,
                          int32_t cap);

/* Host text stage exposed for parity tests (pure CPU, no device): */
/* time_stamp_lfr6_onnx (OfflineRecognizer.cs:200-302): returns count of [begin,end] ms pairs
   written to out_pairs (cap pairs), or a negative status (no fire -> PF_ERR_RECOGNITION). */
int pf_host_timestamps(const float* us_cif_peak, int32_t n, const int64_t* tokens, int32_t n_tokens,
                       int32_t* out_pairs, int32_t cap_pairs);
/* GetHotwords (OfflineRecognizer.cs:72-90) over in-memory token table / hotword lines. */
int pf_host_hotword_ids(const char* const* tokens, int32_t n_tokens, const char* const* lines,
                        int32_t n_lines, int32_t* ids, int32_t ids_cap, int32_t* lens,
                        int32_t lens_cap, int32_t* n_hotwords);
/* DecodeMulti for one stream (OfflineRecognizer.cs:304-418) over an in-memory token table.
   timestamps: flattened ints + per-entry lengths. Result is read with pf_result_* on the
   returned scratch recognizer-less decoder object. */
typedef struct pf_decoded pf_decoded;
int pf_host_decode(const char* const* tokens, int32_t n_tokens, const int64_t* ids, int32_t n_ids,
                   const int32_t* ts_ints, const int32_t* ts_lens, int32_t n_ts, pf_decoded** out);
int pf_decoded_text(pf_decoded* d, const char** utf8, int32_t* text_len_utf16);
int pf_decoded_num_tokens(pf_decoded* d, int32_t* n);
int pf_decoded_token(pf_decoded* d, int32_t j, const char** utf8);
int pf_decoded_num_timestamps(pf_decoded* d, int32_t* n);
int pf_decoded_timestamp(pf_decoded* d, int32_t j, const int32_t** ints, int32_t* n_ints);
void pf_decoded_free(pf_decoded* d);

/* ---- Examples harness helpers (AliParaformerAsr.Examples/Utils/AudioHelper.cs) ---------------------------
   pf_host_wav_read  = GetFileSample (:12-32) for RIFF/WAVE files: decode to float (NAudio AudioFileReader
   conversions), resample + down-mix to 16 kHz mono ONLY when the file's rate is not 16 kHz (upstream quirk: a
   16 kHz stereo file is returned interleaved); a missing file yields one zero sample.  Call with out == NULL to
   learn *n_out.  pf_host_resample = Resample(source, srIn, srOut, channels) (:223-279).
   pf_host_is_audio = IsAudioByHeader (:286-340) restricted to RIFF/WAVE. */
int pf_host_wav_read(const char* path, float* out, int64_t cap, int64_t* n_out, int32_t* sample_rate,
                     int32_t* channels, double* duration_ms);
int pf_host_resample(const float* src, int64_t n, int32_t sr_in, int32_t sr_out, int32_t channels, float* out,
                     int64_t cap, int64_t* n_out);
int pf_host_is_audio(const char* path, int32_t* is_audio);

#ifdef __cplusplus
}
#endif
#endif /* PARAFORMER_HIP_H_ */
