// The helper threads of the recognizer's staged uploads (recognizer.h, CopyLane).  Host-only.
#include "copycrew.h"

#include <algorithm>
#include <chrono>
#include <cstring>

namespace pf {

// ------------------------------------------------------------------ CopyCrew ------------
static inline void cpu_relax() { __builtin_ia32_pause(); }

CopyCrew::CopyCrew(int helpers) {
  try {
    for (int i = 0; i < helpers; ++i) th_.emplace_back([this] { run(); });
  } catch (...) {                                       // no more threads: work with those we have (copy() never needs one)
  }
}

CopyCrew::~CopyCrew() {
  { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
  cv_.notify_all();
  for (auto& t : th_) t.join();
}

bool CopyCrew::take(Job& j) {
  if (queued_.load(std::memory_order_acquire) <= 0) return false;
  std::lock_guard<std::mutex> lk(mu_);
  if (q_.empty()) return false;
  j = q_.front();
  q_.pop_front();
  queued_.fetch_sub(1, std::memory_order_release);
  return true;
}

void CopyCrew::run() {
  using clk = std::chrono::steady_clock;
  for (;;) {
    Job j;
    bool got = false;
    const auto t0 = clk::now();
    for (int spins = 0; !got && !stop_.load(std::memory_order_relaxed); ++spins) {
      got = take(j);
      if (got) break;
      cpu_relax();
      if ((spins & 255) == 255 && clk::now() - t0 > std::chrono::microseconds(400)) break;
    }
    if (!got) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_.load() || !q_.empty(); });
      if (q_.empty()) return;                             // stop
      j = q_.front();
      q_.pop_front();
      queued_.fetch_sub(1, std::memory_order_release);
    }
    std::memcpy(j.d, j.s, j.n);
    j.left->fetch_sub(1, std::memory_order_release);
  }
}

void CopyCrew::copy(char* dst, const char* src, size_t bytes) {
  const size_t kMinShare = (size_t)128 << 10;
  const int parts = (int)std::min<size_t>(th_.size() + 1, bytes / kMinShare);
  if (parts <= 1) { std::memcpy(dst, src, bytes); return; }
  const size_t share = ((bytes / parts) + 4095) & ~(size_t)4095;
  std::atomic<int> left(0);
  size_t mine = std::min(share, bytes);
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (size_t off = mine; off < bytes; off += share) {
      left.fetch_add(1, std::memory_order_relaxed);
      q_.push_back({dst + off, src + off, std::min(share, bytes - off), &left});
      queued_.fetch_add(1, std::memory_order_release);
    }
  }
  cv_.notify_all();
  std::memcpy(dst, src, mine);
  while (left.load(std::memory_order_acquire) > 0) {      // shares nobody has taken yet are ours as well
    Job j;
    if (take(j)) { std::memcpy(j.d, j.s, j.n); j.left->fetch_sub(1, std::memory_order_release); }
    else cpu_relax();
  }
}

}  // namespace pf
