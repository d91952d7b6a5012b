// Mutation fuzz of the host-side parsers that read files a user hands over (csrc/json.h: PFW header / asr.json;
// csrc/hostutil.cpp: asr.yaml, am.mvn, tokens, RIFF/WAVE, resampler, UTF-8), built with -fsanitize=address,undefined
// (tests/test_native_host.py builds and runs it).  Every input either parses or throws pf::Error; nothing else.
//   usage: host_fuzz <iterations> <scratch dir>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "hostutil.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}

static std::string mutate(const std::string& seed) {
  std::string s = seed;
  const int n = 1 + (int)(rnd() % 6);
  for (int i = 0; i < n && !s.empty(); ++i) {
    const size_t pos = rnd() % s.size();
    switch (rnd() % 7) {
      case 0: s[pos] = (char)rnd(); break;
      case 1: s.erase(pos, 1 + rnd() % 8); break;
      case 2: s.insert(pos, 1 + rnd() % 4, (char)rnd()); break;
      case 3: s.resize(pos); break;
      case 4: s.insert(pos, s.substr(rnd() % s.size(), rnd() % 16)); break;
      case 5: { static const char* tok[] = {"[", "{", "\"", "\\u", "\\uD83D", "-", "1e999", ":", ",", "\n", "\r\n", "<", "[ ", " ]", "\xEF\xBB\xBF", "\xF0\x9F"};
                s.insert(pos, tok[rnd() % (sizeof(tok) / sizeof(tok[0]))]); break; }
      default: s[pos] = (char)(s[pos] ^ (1 << (rnd() % 8)));
    }
  }
  return s;
}

template <class F>
static void guarded(F f, long& ok, long& err) {
  try { f(); ++ok; } catch (const pf::Error&) { ++err; }
}

static std::string wav_bytes(int fmt, int bits, int ch, int rate, int frames) {
  std::string d;
  auto u32 = [&](uint32_t v) { d.append((const char*)&v, 4); };
  auto u16 = [&](uint16_t v) { d.append((const char*)&v, 2); };
  const uint32_t data = (uint32_t)frames * ch * bits / 8;
  d += "RIFF"; u32(36 + data); d += "WAVE"; d += "fmt "; u32(16); u16((uint16_t)fmt); u16((uint16_t)ch); u32((uint32_t)rate);
  u32((uint32_t)rate * ch * bits / 8); u16((uint16_t)(ch * bits / 8)); u16((uint16_t)bits); d += "data"; u32(data);
  for (uint32_t i = 0; i < data; ++i) d += (char)rnd();
  return d;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  long ok = 0, err = 0;
  const std::string json_seeds[] = {
      "{\"config\": {\"kind\": \"paraformer\", \"d_model\": 512, \"use_itn\": true, \"x\": null}, \"tensors\": [{\"name\": \"a.weight\", "
      "\"dtype\": \"f32\", \"shape\": [512, 560], \"offset\": 0, \"nbytes\": 1146880}, {\"name\": \"b\", \"dtype\": \"u8\", \"shape\": [4], \"offset\": 256, \"nbytes\": 4}]}",
      "{\"model\": \"SenseVoiceSmall\", \"use_itn\": \"True\", \"frontend_conf\": {\"fs\": 16000, \"window\": \"hamming\", \"n_mels\": 80, "
      "\"frame_length\": 25, \"frame_shift\": 10, \"dither\": 1.0, \"lfr_m\": 7, \"lfr_n\": 6, \"snip_edges\": false}, \"s\": \"\\u4f60\\ud83d\\ude00\\n\"}",
      "[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[1]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]",
      "[1, -2.5e3, 1e400, \"\", {}, [], true, false, null]"};
  const std::string yaml_seed =
      "model: paraformer\nuse_itn: false\nfrontend_conf:\n  fs: 16000\n  window: hamming\n  n_mels: 80\n  frame_length: 25\n"
      "  frame_shift: 10\n  dither: 0.0\n  lfr_m: 7\n  lfr_n: 6\n  snip_edges: false\nencoder_conf:\n  output_size: 512\n# c\n";
  const std::string mvn_seed =
      "<Nnet>\n<Splice> 560 560\n[ 0 ]\n<AddShift> 560 560\n<LearnRateCoef> 0 [ -8.311879 -8.600912 -9.615928 ]\n<Rescale> 560 560\n"
      "<LearnRateCoef> 0 [ 0.155775 0.154484 0.1527379 ]\n</Nnet>\n";
  const std::string utf_seed = "\xE4\xBD\xA0\xE5\xA5\xBD hello \xF0\x9F\x98\x80 \xC3\xA9@@ </s>";
  for (int it = 0; it < iters; ++it) {
    const std::string j = mutate(json_seeds[it % 4]);
    guarded([&] {
      pf::Json v = pf::JsonParser(j.data(), j.size()).parse();
      (void)v.num_or("x", 1); (void)v.str_or("kind", ""); (void)v.bool_or("use_itn", false);
      if (const pf::Json* t = v.get("tensors")) for (const pf::Json& e : t->arr) { (void)e.str_or("name", ""); (void)e.num_or("offset", -1); }
    }, ok, err);
    guarded([&] { (void)pf::conf_from_json(j); }, ok, err);
    const std::string y = mutate(yaml_seed);
    guarded([&] { (void)pf::conf_from_yaml(y); }, ok, err);
    const std::string m = mutate(mvn_seed);
    guarded([&] { std::vector<float> a, b; pf::parse_mvn_text(m, a, b); }, ok, err);
    const std::string u = mutate(utf_seed);
    guarded([&] {
      const std::vector<uint32_t> cps = pf::utf8_decode(u);
      (void)pf::utf8_encode(cps); (void)pf::utf16_length(u); (void)pf::split_lines(u);
    }, ok, err);
    // RIFF/WAVE: a well-formed file of a random format, then a mutated one
    static const int fmts[][2] = {{1, 8}, {1, 16}, {1, 24}, {1, 32}, {3, 32}, {1, 12}, {7, 8}, {0xFFFE, 16}};
    const int* f = fmts[rnd() % 8];
    std::string w = wav_bytes(f[0], f[1], 1 + (int)(rnd() % 3), (rnd() % 4 == 0) ? 16000 : 8000 + (int)(rnd() % 40000), (int)(rnd() % 300));
    if (it % 2) w = mutate(w);
    const std::string path = dir + "/fuzz.wav";
    { std::ofstream o(path, std::ios::binary); o.write(w.data(), (std::streamsize)w.size()); }
    guarded([&] { (void)pf::is_wav_header(path); }, ok, err);
    guarded([&] { pf::WavData d = pf::decode_wav_file(path); (void)d; }, ok, err);
    guarded([&] { double ms = 0; (void)pf::get_file_sample(path, &ms); }, ok, err);
    guarded([&] {
      std::vector<float> src(rnd() % 500);
      for (float& x : src) x = (float)(rnd() % 2000) / 1000.f - 1.f;
      (void)pf::resample_linear(src, 1 + (int)(rnd() % 96000), 16000, 1 + (int)(rnd() % 2));
    }, ok, err);
  }
  std::remove((dir + "/fuzz.wav").c_str());
  std::printf("ok %ld rejected %ld\n", ok, err);
  return 0;
}
