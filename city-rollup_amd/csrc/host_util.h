// Host-side plumbing shared by the entry points: worker threads that cannot take the process down, the fault
// injection behind cp_fault_inject, and the exception -> cp_status mapping of the function-try-blocks around every
// extern "C" body (include/cityprover.h: "NEVER abort or throw"). Host-only, no HIP: tests/hostsim builds it with g++.
#pragma once
#include <atomic>
#include <exception>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

namespace hostu {

// cp_fault_inject: counters of events still to pass before one fails; < 0 = disarmed
inline std::atomic<long> &fault_counter(int kind) {
  static std::atomic<long> c[4] = {{-1}, {-1}, {-1}, {-1}};
  return c[kind & 3];
}
inline bool fault_fires(int kind) {
  std::atomic<long> &c = fault_counter(kind);
  long v = c.load(std::memory_order_relaxed);
  while (v >= 0) {
    if (c.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) return v == 0;
  }
  return false;
}
// a host allocation checkpoint (CP_FAULT_ALLOC = 1)
inline void alloc_checkpoint() {
  if (fault_fires(1)) throw std::bad_alloc();
}

// Start `f` on a new thread. Returns false — and runs nothing — when the thread cannot be created
// (std::system_error from the constructor, or CP_FAULT_THREAD = 0): the caller then does the work itself.
template <class F>
bool try_spawn(std::thread &t, F &&f) noexcept {
  if (fault_fires(0)) return false;
  try {
    t = std::thread(std::forward<F>(f));
    return true;
  } catch (...) {
    return false;
  }
}

// Per-proof host work of a batch: body(p) for every p < n on up to max_threads threads, the calling thread included.
// Indices are handed out through one counter, so whatever number of helpers could be started, every index is done
// exactly once. body must not throw across threads: an exception inside a helper is caught, remembered and rethrown on
// the calling thread after every helper has been joined.
template <class Body>
void parallel_for(size_t n, size_t max_threads, Body body) {
  if (max_threads <= 1 || n <= 1) {
    for (size_t p = 0; p < n; p++) body(p);
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr err;
  std::atomic<bool> failed{false};
  auto run = [&]() noexcept {
    for (;;) {
      const size_t p = next.fetch_add(1, std::memory_order_relaxed);
      if (p >= n || failed.load(std::memory_order_relaxed)) return;
      try {
        body(p);
      } catch (...) {
        if (!failed.exchange(true)) err = std::current_exception();
      }
    }
  };
  std::thread helpers[16];
  size_t started = 0;
  const size_t want = (max_threads > 16 ? 16 : max_threads) - 1;
  for (size_t t = 0; t < want; t++) {
    if (!try_spawn(helpers[started], run)) break;  // carry on with the helpers that did start
    started++;
  }
  run();
  for (size_t t = 0; t < started; t++) helpers[t].join();
  if (failed.load()) std::rethrow_exception(err);
}

// joins on scope exit (an exception between creating a thread and joining it would otherwise be std::terminate)
struct Joiner {
  std::thread t;
  ~Joiner() {
    if (t.joinable()) t.join();
  }
};

}  // namespace hostu
