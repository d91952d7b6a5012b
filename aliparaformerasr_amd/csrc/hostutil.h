// hostutil.h — host-side file/config helpers shared by the engine and the recognizer mirror.
#pragma once
#include <string>
#include <vector>

#include "common.h"
#include "json.h"

namespace pf {

std::string read_text_file(const std::string& path);          // throws PF_ERR_IO
bool file_exists(const std::string& path);
void read_binary_file(const std::string& path, std::vector<char>& out);

// LoadCmvn (AliParaformerAsr/WavFrontend.cs:112-153)
void parse_mvn_text(const std::string& text, std::vector<float>& shift, std::vector<float>& scale);

// ReadTokens (AliParaformerAsr/Utils/PreloadHelper.cs:120-141): File.ReadAllLines semantics
std::vector<std::string> read_lines(const std::string& path);
std::vector<std::string> split_lines(const std::string& text);

// ConfEntity subset consumed on the path (AliParaformerAsr/Model/ConfEntity.cs,
// FrontendConfEntity.cs) with the reference defaults.
struct ConfEntity {
  std::string model = "paraformer";
  bool use_itn = false;
  int fs = 16000;
  std::string window = "hamming";
  int n_mels = 80;
  int frame_length = 25, frame_shift = 10;
  float dither = 1.0f;
  int lfr_m = 7, lfr_n = 6;
  bool snip_edges = false;
};
// LoadConf (AliParaformerAsr/OfflineRecognizer.cs:55-71): ".json" -> json, ".yaml" -> yaml,
// anything else / missing file -> defaults.
ConfEntity load_conf(const std::string& path);
ConfEntity conf_from_yaml(const std::string& text);
ConfEntity conf_from_json(const std::string& text);

// ---- Examples harness (AliParaformerAsr.Examples/Utils/AudioHelper.cs) -------------------------------
// IsAudioByHeader restricted to what this build decodes: RIFF....WAVE in the first 16 bytes (:286-340)
bool is_wav_header(const std::string& path);
// AudioFileReader semantics for RIFF/WAVE (NAudio converts every PCM width to IEEE float): interleaved
// samples, PCM8 -> b/128-1, PCM16 -> /32768, PCM24 -> /8388608, PCM32 -> /2147483648, float32 as is.
struct WavData { std::vector<float> samples; int sample_rate = 0, channels = 0; double duration_ms = 0; };
WavData decode_wav_file(const std::string& path);
// Resample(sourceData, sourceSampleRate, targetSampleRate, sourceChannels) (:223-279): stereo -> mono
// average first, then linear interpolation in double precision; target length = Round(n / ratio) (banker's)
std::vector<float> resample_linear(const std::vector<float>& src, int sr_in, int sr_out, int channels);
// GetFileSample (:12-32): missing file -> float[1]{0}; resampled (and down-mixed) ONLY when the rate is not
// 16 kHz — a 16 kHz stereo file is handed over interleaved, exactly as upstream does.
std::vector<float> get_file_sample(const std::string& path, double* duration_ms);

// UTF-8 <-> code points
std::vector<uint32_t> utf8_decode(const std::string& s);
std::string utf8_encode(uint32_t cp);
std::string utf8_encode(const std::vector<uint32_t>& cps);
int utf16_length(const std::string& utf8);   // C# string.Length

}  // namespace pf
