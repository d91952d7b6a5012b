// Host-only (no HIP): tests/native/copycrew_sanitize.cpp builds this file alone under TSan / ASan.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace pf {

// Helper threads that share the HOST side of a staged upload (caller's array -> pinned ring): one core copies 1.9 MB (30 s of
// samples) in ~130 us, four in a third of that.  Helpers spin for a short while after a job (a batch arrives as a burst of
// AddSamples calls) and sleep on a condition variable otherwise; the calling thread copies a share itself and, while it waits,
// takes queued shares too — a call never depends on a helper being awake.
class CopyCrew {
 public:
  explicit CopyCrew(int helpers);
  ~CopyCrew();
  void copy(char* dst, const char* src, size_t bytes);
 private:
  struct Job { char* d; const char* s; size_t n; std::atomic<int>* left; };
  bool take(Job& j);
  void run();
  std::mutex mu_; std::condition_variable cv_; std::deque<Job> q_;
  std::atomic<int> queued_{0}; std::atomic<bool> stop_{false};
  std::vector<std::thread> th_;
};

}  // namespace pf
