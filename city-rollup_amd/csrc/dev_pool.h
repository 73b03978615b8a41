// Device buffer pool of the batch handles (fri_prove.inc), over an allocator policy so that the logic can be unit-tested on
// the CPU (tests/hostsim: a counting allocator with a capacity). The product instantiates it on hipMalloc / hipFree (core.h).
//
// A commitment is four or five hipMalloc and, at destroy, as many hipFree - each hipFree a device-wide synchronisation that
// stalls whatever the other contexts of the process have in flight. Freed commitment buffers therefore go back to a
// per-device pool (exact size classes: the shapes of a prover repeat) and are handed out again. Only buffers NO stream can
// still be using may enter it (`reusable`): the library synchronises its own stream before every return, and a handle whose
// device pointers were given to the caller (cp_batch_device_ptrs) is freed through the runtime instead - the caller's kernels
// run on streams the library cannot see, and hipFree waits for the whole device. The cap (CITYPROVER_BATCH_POOL_MB, default
// 4 096; 0 turns the pool off) bounds what a device's pool keeps. Every allocation of the library goes through `malloc`, which
// empties the pool and tries again before it reports out-of-memory. A device index outside the table has NO pool (bypass:
// one device's buffer must never be handed to another).
#pragma once
#include <cstddef>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

// Raw: static int malloc(void **p, size_t bytes) (0 = ok, Raw::OOM = out of memory, anything else = another failure);
//      static void free(void *p)
template <class Raw>
class DevPoolT {
 public:
  struct Stats { size_t bytes, buffers, hits, misses, trims; };

  DevPoolT(size_t n_devices, size_t cap_bytes) : pools_(n_devices), cap_(cap_bytes) {}

  bool has_pool(int device) const { return device >= 0 && (size_t)device < pools_.size(); }

  // give everything the pool of `device` holds back to the runtime; returns the bytes released
  size_t trim(int device) {
    if (!has_pool(device)) return 0;
    One &P = pools_[(size_t)device];
    std::vector<void *> drop;
    size_t released;
    {
      std::lock_guard<std::mutex> l(P.m);
      for (auto &kv : P.free) drop.push_back(kv.second);
      P.free.clear();
      released = P.bytes;
      P.bytes = 0;
      if (released) P.trims++;
    }
    for (void *q : drop) Raw::free(q);
    return released;
  }

  // the allocation every part of the library uses: out of memory with buffers parked in the pool is not out of memory
  int malloc(int device, void **p, size_t bytes) {
    int e = Raw::malloc(p, bytes);
    if (e == Raw::OOM && trim(device)) e = Raw::malloc(p, bytes);
    return e;
  }

  // a buffer of exactly `bytes`: from the pool when it holds one, else from the runtime
  int alloc(int device, void **p, size_t bytes) {
    if (has_pool(device)) {
      One &P = pools_[(size_t)device];
      std::lock_guard<std::mutex> l(P.m);
      auto it = P.free.find(bytes);
      if (it != P.free.end()) {
        *p = it->second;
        P.free.erase(it);
        P.bytes -= bytes;
        P.hits++;
        return 0;
      }
      P.misses++;
    }
    return malloc(device, p, bytes);
  }

  // reusable: no stream can still be using the buffer (see above); otherwise Raw::free, which waits for the device
  void release(int device, void *p, size_t bytes, bool reusable) {
    if (!p) return;
    if (reusable && has_pool(device)) {
      One &P = pools_[(size_t)device];
      std::lock_guard<std::mutex> l(P.m);
      if (P.bytes + bytes <= cap_) {
        P.free.emplace(bytes, p);
        P.bytes += bytes;
        return;
      }
    }
    Raw::free(p);
  }

  Stats stats(int device) {
    Stats s{0, 0, 0, 0, 0};
    if (!has_pool(device)) return s;
    One &P = pools_[(size_t)device];
    std::lock_guard<std::mutex> l(P.m);
    s.bytes = P.bytes; s.buffers = P.free.size(); s.hits = P.hits; s.misses = P.misses; s.trims = P.trims;
    return s;
  }
  void set_cap(size_t cap_bytes) { cap_ = cap_bytes; }
  size_t cap() const { return cap_; }

 private:
  struct One {
    std::mutex m;
    std::multimap<size_t, void *> free;
    size_t bytes = 0, hits = 0, misses = 0, trims = 0;
  };
  std::deque<One> pools_;  // deque: elements hold a mutex and never move
  size_t cap_;
};
