// k_frontend.hip — WavFrontend on the device: kaldi fbank, LFR, CMVN, PadSequence.
//
// Reference sites replaced:
//   AliParaformerAsr/WavFrontend.cs:31-37   GetFbank: x*32768 then SpeechFeatures.OnlineFbank
//                                           (kaldi-native-fbank; algorithm restated from the
//                                           published kaldi feature-window / mel-computations)
//   AliParaformerAsr/WavFrontend.cs:73-111  ApplyLfr  (3 ZERO left-context frames, floor count)
//   AliParaformerAsr/WavFrontend.cs:53-71   ApplyCmvn ((x + shift) * scale, two roundings)
//   AliParaformerAsr/Utils/PadHelper.cs:23-65 PadSequence (right pad with 0, then every value
//                                           == 0.0f becomes float32(-23.025850929940457f*32768))
//
// fbank: one wavefront per 25 ms frame, four frames per workgroup.  The 400 samples (mirror-reflected at the utterance
// edges when snip_edges == false; interior frames take a branch-free coalesced path) are staged in LDS; each lane then
// forms its four complex points z[n] = x[2n] + i x[2n+1] of the 512-point real transform (DC removal, pre-emphasis,
// window) in registers and runs a 256-point complex Stockham FFT as FOUR radix-4 passes (one in-lane butterfly per
// pass; twiddles from a workgroup-shared LDS table) instead of eight radix-2 passes, then the real-input split and
// the power spectrum.  The 80 triangular mel filters are stored sparse and cut into chunks of at most eight bins so
// that the 64 lanes share the ~500 multiply-adds evenly (a lane per filter leaves the wide high-frequency filters on
// 16 lanes); chunk sums are combined per filter in a fixed order (deterministic).
// Instruction-issue bound (round 2: ~1300 instructions per frame, 204 us for 96 000 frames), not HBM-bound:
// 4 B/sample in (each sample is touched by 2.5 frames, served from L2), 320 B/frame out.
#include "kernels.h"
#include "exact.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace pf {

#define FB_FRAME_LEN 400
#define FB_SHIFT 160
#define FB_NFFT 512
#define FB_MAX_BINS 128

struct FbankTables {
  int n_mels;
  float* window;    // [400]
  float2* tw512;    // [256]  e^{-2 pi i m / 512}
  int* mel_start;   // [n_mels]
  int* mel_off;     // [n_mels + 1] offsets into mel_w
  float* mel_w;     // packed non-zero weights
  // the same filters cut into chunks of <= 8 consecutive bins: chunk c = (first bin, first weight, count);
  // filter m owns chunks [mel_chunk_off[m], mel_chunk_off[m + 1])
  int n_chunks, n_weights;
  int4* mel_chunk;        // [n_chunks]  (x = first bin, y = offset into mel_w, z = count, w = filter)
  int* mel_chunk_off;     // [n_mels + 1]
};

#define FB_MAX_CHUNKS 192      // three rounds of 64 lanes
#define FB_MAX_WEIGHTS 1024

FbankTables* fbank_tables_create(int n_mels, int fs, const char* window) {
  PF_CHECK(n_mels > 0 && n_mels <= FB_MAX_BINS, PF_ERR_INVALID_ARG, "fbank: n_mels out of range");
  std::string wt = window ? window : "hamming";
  std::vector<float> win(FB_FRAME_LEN);
  const double a = 2.0 * M_PI / (FB_FRAME_LEN - 1);
  for (int i = 0; i < FB_FRAME_LEN; ++i) {
    double w;
    if (wt == "hamming") w = 0.54 - 0.46 * std::cos(a * i);
    else if (wt == "hanning") w = 0.5 - 0.5 * std::cos(a * i);
    else if (wt == "povey") w = std::pow(0.5 - 0.5 * std::cos(a * i), 0.85);
    else if (wt == "rectangular") w = 1.0;
    else throw Error(PF_ERR_UNSUPPORTED, "fbank: unsupported window type " + wt);
    win[i] = (float)w;
  }
  std::vector<float2> tw(256);
  for (int m = 0; m < 256; ++m) {
    const double ang = -2.0 * M_PI * m / 512.0;
    tw[m] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  // kaldi MelBanks (float arithmetic): low 20 Hz, high = Nyquist, bins over FFT bins 0..255
  auto mel = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
  const float nyq = 0.5f * fs;
  const float bin_w = (float)fs / FB_NFFT;
  const float mel_low = mel(20.0f), mel_high = mel(nyq);
  const float delta = (mel_high - mel_low) / (float)(n_mels + 1);
  std::vector<int> start(n_mels), off(n_mels + 1, 0);
  std::vector<float> wts;
  for (int b = 0; b < n_mels; ++b) {
    const float left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
    int first = -1;
    for (int i = 0; i < FB_NFFT / 2; ++i) {
      const float m = mel(bin_w * i);
      if (m > left && m < right) {
        const float w = (m <= center) ? (m - left) / (center - left) : (right - m) / (right - center);
        if (first < 0) first = i;
        wts.push_back(w);
      }
    }
    start[b] = first < 0 ? 0 : first;
    off[b + 1] = (int)wts.size();
  }
  if (wts.empty()) wts.push_back(0.f);
  auto* t = new FbankTables();
  t->n_mels = n_mels;
  PF_HIP(hipMalloc(&t->window, sizeof(float) * FB_FRAME_LEN));
  PF_HIP(hipMalloc(&t->tw512, sizeof(float2) * 256));
  PF_HIP(hipMalloc(&t->mel_start, sizeof(int) * n_mels));
  PF_HIP(hipMalloc(&t->mel_off, sizeof(int) * (n_mels + 1)));
  PF_HIP(hipMalloc(&t->mel_w, sizeof(float) * wts.size()));
  PF_HIP(hipMemcpy(t->window, win.data(), sizeof(float) * FB_FRAME_LEN, hipMemcpyHostToDevice));
  PF_HIP(hipMemcpy(t->tw512, tw.data(), sizeof(float2) * 256, hipMemcpyHostToDevice));
  PF_HIP(hipMemcpy(t->mel_start, start.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice));
  PF_HIP(hipMemcpy(t->mel_off, off.data(), sizeof(int) * (n_mels + 1), hipMemcpyHostToDevice));
  PF_HIP(hipMemcpy(t->mel_w, wts.data(), sizeof(float) * wts.size(), hipMemcpyHostToDevice));
  std::vector<int> choff(n_mels + 1, 0);
  std::vector<int> chunks;                                   // 4 ints per chunk
  for (int b = 0; b < n_mels; ++b) {
    const int cnt = off[b + 1] - off[b];
    for (int c0 = 0; c0 < cnt; c0 += 8) {
      chunks.push_back(start[b] + c0); chunks.push_back(off[b] + c0); chunks.push_back(std::min(8, cnt - c0)); chunks.push_back(b);
    }
    choff[b + 1] = (int)chunks.size() / 4;
  }
  t->n_chunks = (int)chunks.size() / 4;
  t->n_weights = (int)wts.size();
  PF_CHECK(t->n_chunks <= FB_MAX_CHUNKS && t->n_weights <= FB_MAX_WEIGHTS, PF_ERR_UNSUPPORTED, "fbank: mel filter bank too dense for the kernel's tables");
  if (chunks.empty()) chunks.assign(4, 0);
  PF_HIP(hipMalloc(&t->mel_chunk, sizeof(int) * chunks.size()));
  PF_HIP(hipMalloc(&t->mel_chunk_off, sizeof(int) * (n_mels + 1)));
  PF_HIP(hipMemcpy(t->mel_chunk, chunks.data(), sizeof(int) * chunks.size(), hipMemcpyHostToDevice));
  PF_HIP(hipMemcpy(t->mel_chunk_off, choff.data(), sizeof(int) * (n_mels + 1), hipMemcpyHostToDevice));
  return t;
}

void fbank_tables_destroy(FbankTables* t) {
  if (!t) return;
  hipFree(t->window); hipFree(t->tw512); hipFree(t->mel_start); hipFree(t->mel_off); hipFree(t->mel_w);
  hipFree(t->mel_chunk); hipFree(t->mel_chunk_off);
  delete t;
}

// Dither (kaldi feature-window.cc Dither(): window[i] += RandGauss() * dither, per extracted frame window,
// before the DC removal; reference default 1.0, Model/FrontendConfEntity.cs:12).  The reference's draw is not
// reproducible, so the device draws from a counter-based generator: normal(seed, frame, sample) by Box-Muller
// over two 32-bit hashes — stateless (any launch geometry gives the same features for the same seed) and
// independent across frames as kaldi's per-window draw is.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {    // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float gauss_at(uint32_t seed, uint64_t frame, uint32_t i) {
  const uint32_t k = mix32(seed ^ 0x9e3779b9u) ^ mix32((uint32_t)frame * 0x27d4eb2fu + (uint32_t)(frame >> 32));
  const uint32_t a = mix32(k + 2u * i + 0x165667b1u), b = mix32((k ^ 0x5bd1e995u) + 2u * i + 1u);
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);              // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// A frame's LDS buffers are touched by ONE wavefront only, and a wave's DS operations execute in issue order, so the
// passes need no s_barrier — only a fence that keeps the compiler from moving LDS accesses across the pass boundary
// (block-wide barriers made four unrelated frames wait for each other thirteen times per frame).
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

struct FbankDev {
  const float* audio; const int64_t* audio_off; const int64_t* n_samples; const int64_t* frame_off;
  int B; int64_t total_frames; int snip_edges, n_mels;
  const float* window; const float2* tw512; const float* mel_w; const int4* mel_chunk; const int* mel_chunk_off;
  int n_chunks, n_weights;
  float dither; uint32_t dither_seed;
  float* out;
};

// 4 frames per 256-thread block, one wavefront each.  LDS: per wave 2 x 256 complex (4 KiB) + shared tables
// (twiddles 2 x 2 KiB, window 1.6 KiB, mel weights <= 4 KiB, chunk table <= 3 KiB).
__global__ __launch_bounds__(256) void fbank_kernel(FbankDev p) {
  __shared__ float2 lds[4][2][256];
  __shared__ float2 tw256_s[256];        // e^{-2 pi i t / 256}
  __shared__ float2 tw512_s[256];        // e^{-2 pi i t / 512}
  __shared__ float win_s[FB_NFFT];       // window, zero beyond 400
  __shared__ float melw_s[FB_MAX_WEIGHTS];
  __shared__ int4 chunk_s[FB_MAX_CHUNKS];
  __shared__ float part_s[4][FB_MAX_CHUNKS];
  __shared__ int choff_s[FB_MAX_BINS + 1];   // chunk ranges of the filters (was two global loads per filter and frame)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  {                                                     // workgroup-shared tables
    const float2 w = p.tw512[tid];
    tw512_s[tid] = w;
    const float2 w2 = p.tw512[(2 * tid) & 255];         // e^{-2 pi i t / 256} = +-tw512[2t mod 256]
    tw256_s[tid] = tid < 128 ? w2 : make_float2(-w2.x, -w2.y);
    win_s[tid] = p.window[tid];
    win_s[tid + 256] = tid + 256 < FB_FRAME_LEN ? p.window[tid + 256] : 0.f;
    for (int i = tid; i < p.n_weights; i += 256) melw_s[i] = p.mel_w[i];
    for (int i = tid; i < p.n_chunks; i += 256) chunk_s[i] = p.mel_chunk[i];
    for (int i = tid; i <= p.n_mels; i += 256) choff_s[i] = p.mel_chunk_off[i];
  }
  __syncthreads();
  float2* bufA = lds[wv][0];
  float2* bufB = lds[wv][1];
  float* raw = reinterpret_cast<float*>(bufB);           // 512 floats: the raw frame
  // Persistent: a wave walks frames gf, gf + 4 G, ... so that the tables above are filled once per workgroup and the
  // samples of the NEXT frame are requested before the current one is transformed (the kernel was bound by the chain
  // of dependent global loads in front of every frame — five for the utterance search alone — at 20-40 waves per CU,
  // not by instruction issue).  The utterance table lives in registers (lane b holds utterance b) when B <= 64.
  const bool small_b = p.B <= 64;
  const int64_t my_foff = (small_b && lane < p.B) ? p.frame_off[lane] : INT64_MAX;
  const int64_t my_n = (small_b && lane < p.B) ? p.n_samples[lane] : 0;
  const int64_t my_aoff = (small_b && lane < p.B) ? p.audio_off[lane] : 0;
  struct Frame { int64_t n, start; const float* wav; bool interior; };
  auto locate = [&](int64_t gf) __attribute__((always_inline)) -> Frame {
    int b;
    int64_t foff, n, aoff;
    if (small_b) {
      const unsigned long long m = __ballot(my_foff <= gf);           // frame_off ascends from 0: bits 0..b set
      b = 63 - __builtin_clzll(m);
      foff = __shfl(my_foff, b, 64); n = __shfl(my_n, b, 64); aoff = __shfl(my_aoff, b, 64);
    } else {
      int lo = 0, hi = p.B;                              // largest b with frame_off[b] <= gf
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p.frame_off[mid] <= gf) lo = mid; else hi = mid;
      }
      b = lo;
      foff = p.frame_off[b]; n = p.n_samples[b]; aoff = p.audio_off[b];
    }
    Frame fr;
    const int64_t f = gf - foff;
    fr.n = n;
    fr.wav = p.audio + aoff;
    fr.start = p.snip_edges ? f * FB_SHIFT : f * FB_SHIFT + FB_SHIFT / 2 - FB_FRAME_LEN / 2;
    fr.interior = fr.start >= 0 && fr.start + FB_FRAME_LEN <= n;      // wave-uniform
    return fr;
  };
  // raw samples of a frame: lane's elements i = lane + 64 r (x 32768 and dither are applied when they are consumed)
  auto fetch = [&](const Frame& fr, float (&v)[7]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int i = lane + 64 * r;
      v[r] = 0.f;
      if (i < FB_FRAME_LEN) {
        int64_t sidx = fr.start + i;
        if (!fr.interior)
          while (sidx < 0 || sidx >= fr.n) sidx = sidx < 0 ? -sidx - 1 : 2 * fr.n - 1 - sidx;
        v[r] = fr.wav[sidx];
      }
    }
  };
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t gf = (int64_t)blockIdx.x * 4 + wv;               // global frame index
  if (gf >= p.total_frames) return;                        // whole wave; no block-wide barrier follows
  float cur[7], nxt[7];
  fetch(locate(gf), cur);
  for (; gf < p.total_frames; gf += stride) {
  const bool more = gf + stride < p.total_frames;
  if (more) fetch(locate(gf + stride), nxt);               // in flight under this frame's transform

  // ---- x * 32768 (+ dither), partial sums for the DC offset
  float part = 0.f;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int i = lane + 64 * r;
    float v = 0.f;
    if (i < FB_FRAME_LEN) {
      v = cur[r] * 32768.0f;
      if (p.dither != 0.f) v = add_rn(v, mul_rn(gauss_at(p.dither_seed, (uint64_t)gf, (uint32_t)i), p.dither));
      part += v;
    }
    raw[i] = v;
  }
  raw[lane + 448] = 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const float mean = part / (float)FB_FRAME_LEN;
  wave_lds_sync();
  // ---- this lane's four points z[n] = x[2n] + i x[2n+1], n = lane + 64 m: remove DC, pre-emphasis 0.97
  //      (x[i] -= 0.97 x[i-1]; x[0] -= 0.97 x[0]), window; zero beyond sample 399
  float2 u[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i0 = 2 * (lane + 64 * m);
    float2 z = make_float2(0.f, 0.f);
    if (i0 < FB_FRAME_LEN) {
      const float xm = raw[i0 > 0 ? i0 - 1 : 0] - mean, x0 = raw[i0] - mean, x1 = raw[i0 + 1] - mean;
      z.x = sub_rn(x0, mul_rn(0.97f, xm)) * win_s[i0];
      z.y = sub_rn(x1, mul_rn(0.97f, x0)) * win_s[i0 + 1];
    }
    u[m] = z;
  }
  wave_lds_sync();                                       // every read of the raw frame is done: bufB may be overwritten later
  // ---- 256-point complex FFT, Stockham autosort, radix 4: pass Ns takes u_m = src[j + 64 m] * w^(k m),
  //      w = e^{-2 pi i / (4 Ns)}, k = j mod Ns, and writes the butterfly to dst[(j - k) 4 + k + m Ns]
  auto butterfly_store = [&](float2* dst, int Ns, int k) __attribute__((always_inline)) {
    const float2 v0 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y), v1 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
    const float2 v2 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y);
    const float2 d = make_float2(u[1].x - u[3].x, u[1].y - u[3].y);
    const float2 v3 = make_float2(d.y, -d.x);            // -i (u1 - u3)
    const int j0 = ((lane - k) << 2) + k;
    dst[j0] = make_float2(v0.x + v2.x, v0.y + v2.y);
    dst[j0 + Ns] = make_float2(v1.x + v3.x, v1.y + v3.y);
    dst[j0 + 2 * Ns] = make_float2(v0.x - v2.x, v0.y - v2.y);
    dst[j0 + 3 * Ns] = make_float2(v1.x - v3.x, v1.y - v3.y);
  };
  butterfly_store(bufA, 1, 0);                           // Ns = 1: no twiddles
  wave_lds_sync();
  float2* src = bufA;
  float2* dst = bufB;
#pragma unroll
  for (int Ns = 4; Ns < 256; Ns <<= 2) {
    const int k = lane & (Ns - 1);
    const int t = k * (64 / Ns);                         // w^(k m) = tw256[t m], t m <= 189
    u[0] = src[lane];
    u[1] = cmul(src[lane + 64], tw256_s[t]);
    u[2] = cmul(src[lane + 128], tw256_s[2 * t]);
    u[3] = cmul(src[lane + 192], tw256_s[3 * t]);
    wave_lds_sync();
    butterfly_store(dst, Ns, k);
    wave_lds_sync();
    float2* tmp = src; src = dst; dst = tmp;
  }
  // src holds Z[k]; real-input split: X[k] = (Z[k] + conj Z[256-k])/2 - i e^{-2 pi i k/512} (Z[k] - conj Z[256-k])/2
  float* power = reinterpret_cast<float*>(dst);   // 256 floats
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = lane + 64 * r;
    const float2 zk = src[k];
    const float2 zc = src[(256 - k) & 255];
    const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));   // even part
    const float2 o = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));   // (Z - conj Z')/2
    const float2 t = cmul(o, tw512_s[k]);
    // X = e - i*t = (e.x + t.y, e.y - t.x)
    const float re = e.x + t.y, im = e.y - t.x;
    power[k] = re * re + im * im;
  }
  wave_lds_sync();
  // ---- mel energies: chunk sums (<= 8 bins each), then per filter the sum of its chunks in order
  float* parts = part_s[wv];
  for (int c = lane; c < p.n_chunks; c += 64) {
    const int4 ch = chunk_s[c];
    float acc = 0.f;
    for (int i = 0; i < ch.z; ++i) acc += melw_s[ch.y + i] * power[ch.x + i];
    parts[c] = acc;
  }
  wave_lds_sync();
  for (int m = lane; m < p.n_mels; m += 64) {
    const int c0 = choff_s[m], c1 = choff_s[m + 1];
    float acc = 0.f;
    for (int c = c0; c < c1; ++c) acc += parts[c];
    acc = fmaxf(acc, 1.1920929e-07f);
    p.out[gf * p.n_mels + m] = logf(acc);
  }
  wave_lds_sync();                                       // the next frame overwrites this wave's buffers
#pragma unroll
  for (int r = 0; r < 7; ++r) cur[r] = nxt[r];
  }
}

void launch_fbank(hipStream_t s, const FbankTables* tb, const float* audio, const int64_t* audio_off,
                  const int64_t* n_samples, const int64_t* frame_off, int B, int64_t total_frames,
                  int snip_edges, float* fbank, float dither, uint32_t dither_seed) {
  if (total_frames <= 0) return;
  FbankDev p;
  p.audio = audio; p.audio_off = audio_off; p.n_samples = n_samples; p.frame_off = frame_off; p.B = B;
  p.total_frames = total_frames; p.snip_edges = snip_edges; p.n_mels = tb->n_mels;
  p.window = tb->window; p.tw512 = tb->tw512; p.mel_w = tb->mel_w; p.mel_chunk = tb->mel_chunk; p.mel_chunk_off = tb->mel_chunk_off;
  p.n_chunks = tb->n_chunks; p.n_weights = tb->n_weights;
  p.dither = dither; p.dither_seed = dither_seed; p.out = fbank;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    PF_HIP(hipGetDevice(&dev));
    PF_HIP(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
  }
  const int64_t blocks = (total_frames + 3) / 4;
  // the persistent grid = exactly the workgroups that are resident at once (the occupancy the registers and the LDS
  // allow: 4 per CU at 114 VGPRs); a grid of 5 per CU ran its fifth workgroups as a second round: 156 vs 131 us
  static const int per_cu = [] {
    int occ = env_int("PF_FBANK_WGS", 0);
    if (occ <= 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fbank_kernel, 256, 0) != hipSuccess) occ = 4;
    return std::max(1, std::min(occ, 8));
  }();
  hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)std::min<int64_t>(blocks, (int64_t)cus * per_cu)), dim3(256), 0, s, p);
  PF_HIP(hipGetLastError());
}

// -------------------------------------------------------------------------------------------
// LFR + CMVN + right-pad + sentinel, one float4 per thread, written straight into [B,P+Tmax,W].
// P > 0: the P rows of `prompt` [P,W] are prepended to every utterance (SenseVoice query rows,
// OfflineProjOfSenseVoiceSmall.cs:78-106 prepends them to Speech before PadSequence, so the sentinel
// replacement covers them too).
__global__ __launch_bounds__(256) void lfr_cmvn_pad_kernel(const float* __restrict__ fbank,
                                                           const int64_t* __restrict__ frame_off,
                                                           const int32_t* __restrict__ t80, int B, int Tmax,
                                                           int lfr_m, int lfr_n, int n_mels,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ scale, int apply_cmvn,
                                                           int apply_sentinel, const float* __restrict__ prompt,
                                                           int P, float* __restrict__ out) {
  const int W = lfr_m * n_mels;
  const int wq = W >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * (P + Tmax) * wq;
  if (i >= total) return;
  const int qd = (int)(i % wq);
  const int64_t bt = i / wq;
  const int t = (int)(bt % (P + Tmax)) - P;
  const int b = (int)(bt / (P + Tmax));
  const int T80 = t80[b];
  const int T = T80 / lfr_n;                      // floor (WavFrontend.cs:76)
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < 0) {
    v = *reinterpret_cast<const float4*>(prompt + (int64_t)(t + P) * W + qd * 4);
  } else if (t < T) {
    const int j = qd * 4;
    const int sub = j / n_mels, col = j - sub * n_mels;
    int f = t * lfr_n + sub - (lfr_m - 1) / 2;   // left context = ZERO frames (quirk Q1)
    if (f >= T80) f = T80 - 1;                   // tail branch replicates the last frame
    if (f >= 0) v = *reinterpret_cast<const float4*>(fbank + (frame_off[b] + f) * n_mels + col);
    if (apply_cmvn) {
      const float4 sh = *reinterpret_cast<const float4*>(shift + j);
      const float4 sc = *reinterpret_cast<const float4*>(scale + j);
      v.x = mul_rn(add_rn(v.x, sh.x), sc.x);
      v.y = mul_rn(add_rn(v.y, sh.y), sc.y);
      v.z = mul_rn(add_rn(v.z, sh.z), sc.z);
      v.w = mul_rn(add_rn(v.w, sh.w), sc.w);
    }
  }
  if (apply_sentinel) {
    const float S = -23.025850929940457f * 32768.0f;
    v.x = v.x == 0.f ? S : v.x; v.y = v.y == 0.f ? S : v.y;
    v.z = v.z == 0.f ? S : v.z; v.w = v.w == 0.f ? S : v.w;
  }
  reinterpret_cast<float4*>(out)[i] = v;
}

void launch_lfr_cmvn_pad(hipStream_t s, const float* fbank, const int64_t* frame_off, const int32_t* t80, int B,
                         int Tmax, int lfr_m, int lfr_n, int n_mels, const float* shift, const float* scale,
                         int apply_cmvn, int apply_sentinel, float* out, const float* prompt, int P) {
  PF_CHECK(n_mels % 4 == 0, PF_ERR_INVALID_ARG, "lfr: n_mels must be a multiple of 4");
  const int64_t total = (int64_t)B * (P + Tmax) * (lfr_m * n_mels / 4);
  if (total == 0) return;
  hipLaunchKernelGGL(lfr_cmvn_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, fbank,
                     frame_off, t80, B, Tmax, lfr_m, lfr_n, n_mels, shift, scale, apply_cmvn, apply_sentinel, prompt, P,
                     out);
  PF_HIP(hipGetLastError());
}

// ragged [n_floats[b]] feature buffers -> [B, row_floats] right-padded with 0, then sentinel.
__global__ __launch_bounds__(256) void pad_sentinel_kernel(const float* __restrict__ feats,
                                                           const int64_t* __restrict__ feat_off,
                                                           const int32_t* __restrict__ n_floats, int B,
                                                           int row_floats, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * row_floats;
  if (i >= total) return;
  const int b = (int)(i / row_floats);
  const int j = (int)(i - (int64_t)b * row_floats);
  float v = j < n_floats[b] ? feats[feat_off[b] + j] : 0.f;
  const float S = -23.025850929940457f * 32768.0f;
  out[i] = v == 0.f ? S : v;
}

void launch_pad_sentinel(hipStream_t s, const float* feats, const int64_t* feat_off, const int32_t* n_floats,
                         int B, int row_floats, float* out) {
  const int64_t total = (int64_t)B * row_floats;
  if (total == 0) return;
  hipLaunchKernelGGL(pad_sentinel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feats,
                     feat_off, n_floats, B, row_floats, out);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
