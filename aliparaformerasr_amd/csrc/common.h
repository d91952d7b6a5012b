// common.h — shared host-side helpers for libparaformer_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/paraformer_hip.h"

namespace pf {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);

#define PF_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      throw ::pf::Error(PF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define PF_CHECK(cond, code, msg)                     \
  do {                                                \
    if (!(cond)) throw ::pf::Error((code), (msg));    \
  } while (0)

using half_t = _Float16;

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// PF_CU_CAP=n: limits PERSISTENT grid sizes (and their tile-shape rules) to n compute units.  It does not partition the
// device: the hardware still places those workgroups where it likes, and the non-persistent kernels (attention, the
// row-complete and short-input GEMMs) ignore it (tools/: the two-engine overlap experiment of round 4)
// An integer knob read from the environment.  Call sites keep the result in a function-local `static const`, whose
// initialisation C++11 makes thread-safe: launchers run concurrently from the recognizer pool's caller threads.
inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; }
inline int cu_limit(int cus) {
  static const int cap = [] { const char* e = getenv("PF_CU_CAP"); return e ? atoi(e) : 0; }();   // thread-safe initialiser
  return cap > 0 && cap < cus ? cap : cus;
}

}  // namespace pf
