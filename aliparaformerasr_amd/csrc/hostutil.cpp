// hostutil.cpp — see hostutil.h
#include "hostutil.h"

#include <cerrno>
#include <climits>

#include <cmath>
#include <cstring>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pf {

bool file_exists(const std::string& path) {
  if (path.empty()) return false;
  std::ifstream f(path, std::ios::binary);
  return f.good();
}

std::string read_text_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) throw Error(PF_ERR_IO, "cannot open file: " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

void read_binary_file(const std::string& path, std::vector<char>& out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f.good()) throw Error(PF_ERR_IO, "cannot open file: " + path);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  out.resize((size_t)n);
  if (n > 0 && !f.read(out.data(), n)) throw Error(PF_ERR_IO, "short read: " + path);
}

std::vector<std::string> split_lines(const std::string& text) {
  // File.ReadAllLines: splits on \n, \r\n, \r; a trailing newline does not create an empty line
  std::vector<std::string> lines;
  size_t i = 0, n = text.size();
  // skip UTF-8 BOM like StreamReader does
  if (n >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) i = 3;
  std::string cur;
  bool any = false;
  for (; i < n; ++i) {
    char c = text[i];
    if (c == '\n' || c == '\r') {
      lines.push_back(cur);
      cur.clear();
      any = false;
      if (c == '\r' && i + 1 < n && text[i + 1] == '\n') ++i;
    } else {
      cur += c;
      any = true;
    }
  }
  if (any) lines.push_back(cur);
  return lines;
}

std::vector<std::string> read_lines(const std::string& path) { return split_lines(read_text_file(path)); }

static std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
  return s.substr(a, b - a);
}

static bool starts_with(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }

static void parse_bracket_floats(const std::string& line, std::vector<float>& out) {
  const size_t a = line.find('['), b = line.rfind(']');
  if (a == std::string::npos || b == std::string::npos || b <= a)
    throw Error(PF_ERR_FORMAT, "am.mvn: missing [ ] in <LearnRateCoef> line");
  const std::string inner = line.substr(a + 1, b - a - 1);
  out.clear();
  size_t i = 0;
  while (i <= inner.size()) {
    size_t j = inner.find(' ', i);
    if (j == std::string::npos) j = inner.size();
    std::string tok = trim(inner.substr(i, j - i));
    if (!tok.empty()) {
      char* end = nullptr;
      const float v = std::strtof(tok.c_str(), &end);
      if (end == tok.c_str() || *end != '\0') throw Error(PF_ERR_FORMAT, "am.mvn: bad number '" + tok + "'");
      out.push_back(v);
    }
    i = j + 1;
  }
}

void parse_mvn_text(const std::string& text, std::vector<float>& shift, std::vector<float>& scale) {
  shift.clear();
  scale.clear();
  int state = 0;
  for (const std::string& line : split_lines(text)) {
    if (line.empty()) continue;
    if (starts_with(line, "<AddShift>")) { state = 1; continue; }
    if (starts_with(line, "<Rescale>")) { state = 2; continue; }
    if (starts_with(line, "<LearnRateCoef>") && state == 1) { parse_bracket_floats(line, shift); continue; }
    if (starts_with(line, "<LearnRateCoef>") && state == 2) { parse_bracket_floats(line, scale); continue; }
  }
}

// ------------------------------------------------------------------ config ----------------
static std::string unquote(std::string v) {
  v = trim(v);
  if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\'')))
    v = v.substr(1, v.size() - 2);
  return v;
}
static bool to_bool(const std::string& v, bool d) {
  std::string s;
  for (char c : v) s += (char)std::tolower((unsigned char)c);
  if (s == "true" || s == "yes" || s == "on" || s == "1") return true;
  if (s == "false" || s == "no" || s == "off" || s == "0") return false;
  return d;
}

// int.Parse-like for the yaml scalars: out-of-range text is a format error (atoi's behaviour there is undefined)
static int yaml_int(const std::string& key, const std::string& v) {
  errno = 0;
  char* end = nullptr;
  const long long x = std::strtoll(v.c_str(), &end, 10);
  if (end == v.c_str() || errno == ERANGE || x < INT32_MIN || x > INT32_MAX)
    throw Error(PF_ERR_FORMAT, "asr.yaml: '" + key + "' is not an integer: '" + v + "'");
  return (int)x;
}

ConfEntity conf_from_yaml(const std::string& text) {
  ConfEntity c;
  std::string section;   // current top-level mapping key
  for (std::string raw : split_lines(text)) {
    // strip comments (outside quotes)
    bool inq = false;
    char qc = 0;
    for (size_t i = 0; i < raw.size(); ++i) {
      if ((raw[i] == '"' || raw[i] == '\'') && (!inq || raw[i] == qc)) { inq = !inq; qc = raw[i]; }
      if (raw[i] == '#' && !inq && (i == 0 || raw[i - 1] == ' ' || raw[i - 1] == '\t')) { raw = raw.substr(0, i); break; }
    }
    if (trim(raw).empty()) continue;
    size_t indent = 0;
    while (indent < raw.size() && raw[indent] == ' ') ++indent;
    const std::string body = trim(raw);
    const size_t colon = body.find(':');
    if (colon == std::string::npos) continue;
    const std::string key = trim(body.substr(0, colon));
    const std::string val = unquote(body.substr(colon + 1));
    if (indent == 0) {
      section = val.empty() ? key : "";
      if (key == "model" && !val.empty()) c.model = val;
      else if (key == "use_itn") c.use_itn = to_bool(val, c.use_itn);
      continue;
    }
    if (section == "frontend_conf") {
      if (key == "fs") c.fs = yaml_int(key, val);
      else if (key == "window") c.window = val;
      else if (key == "n_mels") c.n_mels = yaml_int(key, val);
      else if (key == "frame_length") c.frame_length = yaml_int(key, val);
      else if (key == "frame_shift") c.frame_shift = yaml_int(key, val);
      else if (key == "dither") c.dither = std::strtof(val.c_str(), nullptr);
      else if (key == "lfr_m") c.lfr_m = yaml_int(key, val);
      else if (key == "lfr_n") c.lfr_n = yaml_int(key, val);
      else if (key == "snip_edges") c.snip_edges = to_bool(val, c.snip_edges);
    }
  }
  return c;
}

ConfEntity conf_from_json(const std::string& text) {
  ConfEntity c;
  Json j = JsonParser(text.data(), text.size()).parse();
  c.model = j.str_or("model", c.model);
  c.use_itn = j.bool_or("use_itn", c.use_itn);
  if (const Json* f = j.get("frontend_conf")) {
    c.fs = (int)f->int_or("fs", c.fs, INT32_MIN, INT32_MAX);
    c.window = f->str_or("window", c.window);
    c.n_mels = (int)f->int_or("n_mels", c.n_mels, INT32_MIN, INT32_MAX);
    c.frame_length = (int)f->int_or("frame_length", c.frame_length, INT32_MIN, INT32_MAX);
    c.frame_shift = (int)f->int_or("frame_shift", c.frame_shift, INT32_MIN, INT32_MAX);
    c.dither = (float)f->num_or("dither", c.dither);
    c.lfr_m = (int)f->int_or("lfr_m", c.lfr_m, INT32_MIN, INT32_MAX);
    c.lfr_n = (int)f->int_or("lfr_n", c.lfr_n, INT32_MIN, INT32_MAX);
    c.snip_edges = f->bool_or("snip_edges", c.snip_edges);
  }
  return c;
}

static std::string lower(std::string s) {
  for (char& ch : s) ch = (char)std::tolower((unsigned char)ch);
  return s;
}
static bool ends_with(const std::string& s, const char* suf) {
  const size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

ConfEntity load_conf(const std::string& path) {
  ConfEntity c;
  if (path.empty()) return c;
  const std::string lp = lower(path);
  if (ends_with(lp, ".json")) {
    if (file_exists(path)) c = conf_from_json(read_text_file(path));
  } else if (ends_with(lp, ".yaml")) {
    if (file_exists(path)) c = conf_from_yaml(read_text_file(path));
  }
  return c;
}

// ------------------------------------------------------------------ UTF-8 -----------------
std::vector<uint32_t> utf8_decode(const std::string& s) {
  std::vector<uint32_t> out;
  size_t i = 0, n = s.size();
  while (i < n) {
    unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    int extra;
    if (c < 0x80) { cp = c; extra = 0; }
    else if ((c >> 5) == 0x6) { cp = c & 0x1F; extra = 1; }
    else if ((c >> 4) == 0xE) { cp = c & 0x0F; extra = 2; }
    else if ((c >> 3) == 0x1E) { cp = c & 0x07; extra = 3; }
    else { cp = 0xFFFD; extra = 0; }
    ++i;
    for (int k = 0; k < extra && i < n; ++k, ++i) cp = (cp << 6) | ((unsigned char)s[i] & 0x3F);
    out.push_back(cp);
  }
  return out;
}

std::string utf8_encode(uint32_t cp) {
  std::string o;
  if (cp < 0x80) o += (char)cp;
  else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
  else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  return o;
}

std::string utf8_encode(const std::vector<uint32_t>& cps) {
  std::string o;
  for (uint32_t c : cps) o += utf8_encode(c);
  return o;
}

int utf16_length(const std::string& utf8) {
  int n = 0;
  for (uint32_t cp : utf8_decode(utf8)) n += cp >= 0x10000 ? 2 : 1;
  return n;
}

// ------------------------------------------------------------------ Examples harness -----
bool is_wav_header(const std::string& path) {
  std::vector<char> b;
  if (!file_exists(path)) return false;
  read_binary_file(path, b);
  if (b.size() < 16) return false;
  return std::memcmp(b.data(), "RIFF", 4) == 0 && std::memcmp(b.data() + 8, "WAVE", 4) == 0;
}

static uint32_t rd32(const char* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
static uint16_t rd16(const char* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

WavData decode_wav_file(const std::string& path) {
  std::vector<char> b;
  read_binary_file(path, b);
  PF_CHECK(b.size() >= 12 && std::memcmp(b.data(), "RIFF", 4) == 0 && std::memcmp(b.data() + 8, "WAVE", 4) == 0,
           PF_ERR_FORMAT, "not a RIFF/WAVE file: " + path);
  size_t pos = 12;
  int fmt_tag = 0, bits = 0, block_align = 0;
  WavData w;
  const char* data = nullptr;
  size_t data_bytes = 0;
  while (pos + 8 <= b.size()) {
    const uint32_t sz = rd32(b.data() + pos + 4);
    const char* body = b.data() + pos + 8;
    const size_t avail = b.size() - (pos + 8);
    if (std::memcmp(b.data() + pos, "fmt ", 4) == 0 && sz >= 16 && avail >= 16) {
      fmt_tag = rd16(body); w.channels = rd16(body + 2); w.sample_rate = (int)rd32(body + 4);
      block_align = rd16(body + 12); bits = rd16(body + 14);
      if (fmt_tag == 0xFFFE && sz >= 26 && avail >= 26) fmt_tag = rd16(body + 24);   // WAVE_FORMAT_EXTENSIBLE sub-format
    } else if (std::memcmp(b.data() + pos, "data", 4) == 0) {
      data = body; data_bytes = std::min<size_t>(sz, avail);
      break;
    }
    pos += 8 + (size_t)sz + (sz & 1);
  }
  PF_CHECK(data && w.channels > 0 && w.sample_rate > 0 && block_align > 0, PF_ERR_FORMAT, "wav: missing fmt/data chunk: " + path);
  const int bps = bits / 8;
  // PCM 8 / 16 / 24 / 32, IEEE float 32 / 64, G.711 A-law (6) and mu-law (7): what NAudio's AudioFileReader turns into float samples
  // for a RIFF/WAVE file (the G.711 forms through a codec that expands them to 16-bit PCM first)
  PF_CHECK((fmt_tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (fmt_tag == 3 && (bits == 32 || bits == 64)) ||
               ((fmt_tag == 6 || fmt_tag == 7) && bits == 8),
           PF_ERR_UNSUPPORTED, "wav: unsupported sample format");
  const size_t n = data_bytes / bps;
  w.samples.resize(n);
  const unsigned char* d = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) {
    const unsigned char* q = d + i * bps;
    float v;
    if (fmt_tag == 7) {                                  // ITU-T G.711 mu-law: ~byte = sign | exponent (3) | mantissa (4)
      const int u = (~q[0]) & 0xFF;
      const int mag = ((((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7)) - 0x84;
      v = (float)((u & 0x80) ? -mag : mag) / 32768.0f;
    } else if (fmt_tag == 6) {                           // A-law: byte ^ 0x55, sign bit set = positive
      const int a = q[0] ^ 0x55, e = (a >> 4) & 7, m = a & 0x0F;
      const int mag = e == 0 ? (m << 4) + 8 : ((m << 4) + 0x108) << (e - 1);
      v = (float)((a & 0x80) ? mag : -mag) / 32768.0f;
    } else if (fmt_tag == 3 && bits == 64) { double d; std::memcpy(&d, q, 8); v = (float)d; }
    else if (fmt_tag == 3) { std::memcpy(&v, q, 4); }
    else if (bits == 16) { int16_t x; std::memcpy(&x, q, 2); v = x / 32768.0f; }
    else if (bits == 24) { int32_t x = (int32_t)((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)(int8_t)q[2] << 16)); v = x / 8388608.0f; }
    else if (bits == 32) { int32_t x; std::memcpy(&x, q, 4); v = x / 2147483648.0f; }
    else { v = q[0] / 128.0f - 1.0f; }
    w.samples[i] = v;
  }
  w.duration_ms = (double)(data_bytes / block_align) * 1000.0 / w.sample_rate;
  return w;
}

std::vector<float> resample_linear(const std::vector<float>& src, int sr_in, int sr_out, int channels) {
  PF_CHECK(sr_in > 0 && sr_out > 0, PF_ERR_INVALID_ARG, "resample: sample rates must be positive");
  PF_CHECK(channels == 1 || channels == 2, PF_ERR_INVALID_ARG, "resample: only 1 or 2 channels");
  if (src.empty()) return {};
  std::vector<float> mono_buf;
  const std::vector<float>* mono = &src;
  if (channels == 2) {
    mono_buf.resize(src.size() / 2);
    for (size_t i = 0; i < mono_buf.size(); ++i) mono_buf[i] = (src[2 * i] + src[2 * i + 1]) * 0.5f;
    mono = &mono_buf;
  }
  const std::vector<float>& m = *mono;
  const double ratio = (double)sr_in / sr_out;
  const int n_out = (int)std::nearbyint((double)m.size() / ratio);       // Math.Round: half to even
  std::vector<float> out((size_t)std::max(n_out, 0));
  const int last = (int)m.size() - 1;
  for (int i = 0; i < n_out; ++i) {
    const double pos = i * ratio;
    const int idx = (int)pos;
    const double fr = pos - idx;
    if (idx >= last) { out[i] = m[last < 0 ? 0 : last]; continue; }
    out[i] = (float)((1 - fr) * m[idx] + fr * m[idx + 1]);
  }
  return out;
}

std::vector<float> get_file_sample(const std::string& path, double* duration_ms) {
  if (!file_exists(path)) return std::vector<float>(1, 0.f);
  WavData w = decode_wav_file(path);
  if (duration_ms) *duration_ms = w.duration_ms;
  if (w.sample_rate != 16000) return resample_linear(w.samples, w.sample_rate, 16000, w.channels);
  return w.samples;
}

}  // namespace pf
