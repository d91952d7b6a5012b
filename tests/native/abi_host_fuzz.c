/* Random-input driver for the host-only entry points of the C-ABI (include/paraformer_hip.h: pf_host_*), linked against
   a copy of the library built with -fsanitize=address,undefined (tools/sanitize_host.sh).  Every call must return a
   status — never crash, never read or write outside what its arguments describe.
     usage: abi_host_fuzz <iterations> */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "paraformer_hip.h"

static uint64_t st = 0x2545F4914F6CDD1Dull;
static uint32_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 32); }
static float frand(void) { return (float)(rnd() % 20001) / 10000.f - 1.f; }

static const char* kTokens[] = {"<blank>", "<s>", "</s>", "a", "b@@", "c", "\xE4\xBD\xA0", "\xE5\xA5\xBD", "hello", "wor@@", "ld", "<unk>",
                                "\xF0\x9F\x98\x80", "", "@@", "x y", "'", "<|zh|>", "<|NEUTRAL|>", "<|Speech|>", "<|woitn|>", "\xE2\x96\x81the"};
#define NTOK ((int)(sizeof(kTokens) / sizeof(kTokens[0])))

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  long calls = 0, errors = 0;
  for (int it = 0; it < iters; ++it) {
    int rc;
    /* ---- timestamps */
    {
      const int n = (int)(rnd() % 400), nt = (int)(rnd() % 24);
      float* peak = (float*)malloc(sizeof(float) * (size_t)(n + 1));
      int64_t* tk = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nt + 1));
      for (int i = 0; i < n; ++i) peak[i] = (rnd() % 9 == 0) ? 1.0f + 0.2f * frand() : 0.3f * fabsf(frand());
      if (n && rnd() % 16 == 0) peak[rnd() % n] = NAN;
      for (int i = 0; i < nt; ++i) tk[i] = (int64_t)(rnd() % 40) - 4;
      const int cap = (int)(rnd() % 40);
      int32_t* pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(cap + 1));
      rc = pf_host_timestamps(peak, n, tk, nt, pairs, cap);
      ++calls; errors += rc < 0;
      free(peak); free(tk); free(pairs);
    }
    /* ---- hot words, decode */
    {
      const char* lines[6];
      char buf[6][48];
      const int nl = (int)(rnd() % 6);
      for (int i = 0; i < nl; ++i) {
        const int len = (int)(rnd() % 40);
        for (int c = 0; c < len; ++c) buf[i][c] = (char)(rnd() % 5 == 0 ? (0x80 | (rnd() & 0x7F)) : ("abc xyhelo\xE4\xBD\xA0"[rnd() % 13]));
        buf[i][len] = 0;
        lines[i] = buf[i];
      }
      int32_t ids[64], lens[8], nh = -1;
      rc = pf_host_hotword_ids(kTokens, NTOK, lines, nl, ids, (int)(rnd() % 64), lens, (int)(rnd() % 8), &nh);
      ++calls; errors += rc != 0;
      const int nid = (int)(rnd() % 40);
      int64_t idv[40];
      for (int i = 0; i < nid; ++i) idv[i] = (int64_t)(rnd() % (NTOK + 6)) - 3;
      int32_t ts_ints[80], ts_lens[40];
      const int nts = rnd() % 3 ? 0 : (int)(rnd() % (nid + 1));
      for (int i = 0; i < nts; ++i) { ts_lens[i] = 2; ts_ints[2 * i] = (int)(rnd() % 30000); ts_ints[2 * i + 1] = (int)(rnd() % 30000); }
      pf_decoded* d = NULL;
      rc = pf_host_decode(kTokens, NTOK, idv, nid, ts_ints, ts_lens, nts, &d);
      ++calls; errors += rc != 0;
      if (rc == 0 && d) {
        const char* s = NULL; int32_t l16 = 0, n = 0;
        pf_decoded_text(d, &s, &l16);
        if (s) (void)strlen(s);
        pf_decoded_num_tokens(d, &n);
        for (int j = -1; j <= n; ++j) { const char* t = NULL; if (pf_decoded_token(d, j, &t) == 0 && t) (void)strlen(t); }
        pf_decoded_num_timestamps(d, &n);
        for (int j = -1; j <= n; ++j) { const int32_t* p = NULL; int32_t k = 0; if (pf_decoded_timestamp(d, j, &p, &k) == 0) for (int q = 0; q < k; ++q) (void)p[q]; }
        pf_decoded_free(d);
      }
      char out[64];
      rc = pf_host_online_decode(kTokens, NTOK, idv, nid, out, (int)(rnd() % 64));
      ++calls; errors += rc < 0;
    }
    /* ---- streaming host pieces */
    {
      const int t80 = (int)(rnd() % 60), m = 1 + (int)(rnd() % 9), nn = 1 + (int)(rnd() % 8);
      float* fb = (float*)malloc(sizeof(float) * 80 * (size_t)(t80 + 1));
      for (int i = 0; i < 80 * t80; ++i) fb[i] = frand();
      const int64_t cap = (int64_t)(rnd() % 20) * 80 * m;
      float* out = (float*)malloc(sizeof(float) * (size_t)(cap + 1));
      int32_t tl = -1;
      rc = pf_host_online_lfr(fb, t80, m, nn, out, cap, &tl);
      ++calls; errors += rc != 0;
      free(fb); free(out);
      const int T = (int)(rnd() % 30), D = 2 * (int)(rnd() % 40);
      float* x = (float*)malloc(sizeof(float) * (size_t)(T * D + 1));
      for (int i = 0; i < T * D; ++i) x[i] = frand();
      rc = pf_host_online_posenc(x, T, D, (int)(rnd() % 5000) - 10);
      ++calls; errors += rc != 0;
      free(x);
      const int n = (int)(rnd() % 40);
      float* al = (float*)malloc(sizeof(float) * (size_t)(n + 1));
      for (int i = 0; i < n; ++i) al[i] = fabsf(frand());
      rc = pf_host_online_dynamic_mask(al, n);
      ++calls; errors += rc != 0;
      const int Dh = 1 + (int)(rnd() % 16), fc = (int)(rnd() % 12);
      float* hid = (float*)malloc(sizeof(float) * (size_t)(n * Dh + 1));
      for (int i = 0; i < n * Dh; ++i) hid[i] = frand();
      float* fired = (float*)malloc(sizeof(float) * (size_t)(fc * Dh + 1));
      float ca = fabsf(frand()) * 0.9f;
      float* ch = (float*)calloc((size_t)Dh, sizeof(float));
      int32_t nf = -1;
      rc = pf_host_online_cif(hid, al, n, Dh, 1.0f, fired, fc, &nf, &ca, ch);
      ++calls; errors += rc != 0;
      free(al); free(hid); free(fired); free(ch);
    }
    /* ---- resampler, shard runner */
    {
      const int chn = 1 + (int)(rnd() % 2);
      const int64_t n = (int64_t)(rnd() % 300) * chn;
      float* src = (float*)malloc(sizeof(float) * (size_t)(n + 1));
      for (int64_t i = 0; i < n; ++i) src[i] = frand();
      int64_t no = -1;
      const int sr = (rnd() % 8 == 0) ? (int)(rnd() % 3) - 1 : 4000 + (int)(rnd() % 60000);
      rc = pf_host_resample(src, n, sr, 16000, chn, NULL, 0, &no);
      ++calls; errors += rc != 0;
      if (rc == 0 && no >= 0) {
        const int64_t cap = rnd() % 4 ? no : no / 2;
        float* out = (float*)malloc(sizeof(float) * (size_t)(cap + 1));
        rc = pf_host_resample(src, n, sr, 16000, chn, out, cap, &no);
        ++calls; errors += rc != 0;
        free(out);
      }
      free(src);
      if (it % 8 == 0) {
        const int G = 1 + (int)(rnd() % 6), B = (int)(rnd() % 20);
        int32_t fire[20], tn[20], L = -1;
        for (int i = 0; i < B; ++i) fire[i] = 1 + (int)(rnd() % 9);
        int64_t ids[20 * 9];
        rc = pf_host_group_sim(G, B, fire, (int)(rnd() % 2), 5, (int)(rnd() % 2), (int)(rnd() % (G + 1)) - 1, (int)(rnd() % 3), ids,
                               (int)(rnd() % 4 ? 9 : 3), tn, &L);
        ++calls; errors += rc != 0;
      }
    }
  }
  printf("calls %ld with-error-status %ld\n", calls, errors);
  return 0;
}
