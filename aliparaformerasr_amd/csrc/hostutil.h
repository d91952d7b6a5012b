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

// UTF-8 <-> code points
std::vector<uint32_t> utf8_decode(const std::string& s);
std::string utf8_encode(uint32_t cp);
std::string utf8_encode(const std::vector<uint32_t>& cps);
int utf16_length(const std::string& utf8);   // C# string.Length

}  // namespace pf
