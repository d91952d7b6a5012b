// Sanitizer harness for the copy helpers of the recognizer's staged uploads (csrc/copycrew.cpp): several caller threads copy
// buffers of many sizes through ONE crew at the same time (each call splits into shares, helpers and callers take them), crews are
// created and destroyed while idle, spinning and asleep.  Built with -fsanitize=thread / address by tests/test_native_host.py.
//   usage: copycrew_sanitize <iterations>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "copycrew.h"

static unsigned char pat(size_t i, unsigned salt) { return (unsigned char)((i * 131u + salt * 29u + (i >> 9)) & 0xff); }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 50;
  const size_t sizes[] = {0, 1, 4095, 131072, 262144 + 7, 1000003, 1920000, 2 << 20, (3 << 20) + 123};
  long copies = 0;
  for (int it = 0; it < iters; ++it) {
    const int helpers = it % 5;                                       // 0 helpers: the calling thread copies everything
    pf::CopyCrew crew(helpers);
    if (it % 3 == 1) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // helpers asleep on the condition variable
    const int callers = 1 + it % 4;
    std::vector<std::thread> th;
    std::vector<int> bad(callers, 0);
    for (int c = 0; c < callers; ++c)
      th.emplace_back([&, c] {
        for (int k = 0; k < 6; ++k) {
          const size_t n = sizes[(it + c * 3 + k) % (sizeof(sizes) / sizeof(sizes[0]))];
          std::vector<char> src(n + 1), dst(n + 2, (char)0x5a);
          for (size_t i = 0; i < n; ++i) src[i] = (char)pat(i, (unsigned)(c + k));
          crew.copy(dst.data() + 1, src.data(), n);
          if (n && std::memcmp(dst.data() + 1, src.data(), n) != 0) bad[c] = 1;
          if (dst[0] != (char)0x5a || dst[n + 1] != (char)0x5a) bad[c] = 2;       // nothing outside [dst, dst + n)
        }
      });
    for (auto& t : th) t.join();
    for (int c = 0; c < callers; ++c)
      if (bad[c]) { std::printf("caller %d of iteration %d: bad copy (%d)\n", c, it, bad[c]); return 2; }
    copies += callers * 6;
  }
  std::printf("ok %ld copies\n", copies);
  return 0;
}
