// The host logic of cp_batcher (city-rollup_amd/csrc/batcher.inc — the product source, included below) under
// ThreadSanitizer, without a GPU: the proving call it merges requests into is replaced by a stand-in that sleeps, checks
// what a real batch would check (one shape per batch, one batch per slot at a time) and returns bytes derived from the
// request, so that every caller can verify it received ITS result. Test scaffolding only (tests/test_batcher_sim.py).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cityprover.h"

// ---- the few internals of the library batcher.inc touches, reduced to what it reads ---------------------------------------
struct cp_ctx {
  cp_ctx *parent = nullptr;
  int n_lanes = 1;
  std::vector<cp_ctx *> lanes;
  std::string error;
  std::atomic<int> busy{0};  // batches running on this slot right now: must never exceed 1
};
namespace quot { struct Gate { int type, a, b; }; }
struct cp_circuit {
  cp_ctx *ctx;
  cp_shape sh;
  int num_selectors = 1;
  std::vector<quot::Gate> gates;
  bool has_gates = true;
  int id;
};
namespace {
thread_local std::string g_tls_error;
int set_error(cp_ctx *ctx, int code, const char *fmt, ...) noexcept {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  try { g_tls_error = buf; if (ctx) ctx->error = buf; } catch (...) {}
  return code;
}
int exception_status(cp_ctx *ctx) noexcept { return set_error(ctx, CP_ERR_INTERNAL, "internal error"); }
#define CP_CATCH(ctxexpr) catch (...) { return exception_status(ctxexpr); }
bool same_shape(const cp_shape &a, const cp_shape &b) { return memcmp(&a, &b, sizeof(cp_shape)) == 0; }
size_t max_batch(const cp_shape &sh) { return sh.degree_bits == 13 ? 3 : 4096; }  // the second shape fits three proofs per launch
// the library's argument checks (prover_tail.inc): what they refuse never reaches a batch; the stand-in accepts everything,
// so that the 0xBAD requests below still exercise the fail-alone / retry-singly path of a failing BATCH
int validate_batch(cp_ctx *, size_t, cp_circuit *const *, const uint64_t *const *, const size_t *, bool, const int *, const uint64_t *,
                   std::vector<int> *, std::vector<uint64_t> *, const cp_ctx * = nullptr) { return CP_OK; }
std::atomic<int> g_violations{0};
std::atomic<long> g_batches{0};
}  // namespace

extern "C" const char *cp_last_error(cp_ctx *ctx) { return ctx ? ctx->error.c_str() : g_tls_error.c_str(); }

// stand-in for the GPU prover: proof = 16 bytes (circuit id, first wire value); a wire value of 0xBAD fails the request —
// and, like the real call, the whole batch it is part of
extern "C" int cp_prove_batch_host(cp_ctx *ctx, size_t n, cp_circuit *const *circuits, const uint64_t *const *pis, const size_t *npi,
                                   const uint64_t *const *wires, const int *, const uint64_t *, uint8_t **out, size_t *lens) {
  if (ctx->busy.fetch_add(1) != 0) g_violations++;
  g_batches++;
  int rc = CP_OK;
  for (size_t i = 0; i < n; i++) {
    if (!same_shape(circuits[i]->sh, circuits[0]->sh)) { g_violations++; rc = set_error(ctx, CP_ERR_INVALID_ARG, "mixed shapes"); }
    if (n > max_batch(circuits[0]->sh)) { g_violations++; rc = set_error(ctx, CP_ERR_INVALID_ARG, "batch too large for this shape"); }
    if (wires[i][0] == 0xBAD) rc = set_error(ctx, CP_ERR_INVALID_ARG, "request with wire 0xBAD is not canonical");
    (void)pis; (void)npi;
  }
  std::this_thread::sleep_for(std::chrono::microseconds(200 + 20 * n));
  for (size_t i = 0; i < n; i++) { out[i] = nullptr; lens[i] = 0; }
  if (rc == CP_OK)
    for (size_t i = 0; i < n; i++) {
      out[i] = (uint8_t *)malloc(16);
      const uint64_t id = (uint64_t)circuits[i]->id;
      memcpy(out[i], &id, 8);
      memcpy(out[i] + 8, &wires[i][0], 8);
      lens[i] = 16;
    }
  ctx->busy.fetch_sub(1);
  return rc;
}

#include "../../city-rollup_amd/csrc/batcher.inc"

// usage: batcher_sim <lanes> <threads> <calls per thread> <max_batch> <linger_us>
int main(int argc, char **argv) {
  if (argc != 6) return 2;
  const int lanes = atoi(argv[1]), threads = atoi(argv[2]), calls = atoi(argv[3]), max_batch = atoi(argv[4]), linger = atoi(argv[5]);
  cp_ctx ctx;
  std::vector<cp_ctx> lane_ctx((size_t)lanes);
  ctx.n_lanes = lanes;
  for (auto &l : lane_ctx) { l.parent = &ctx; ctx.lanes.push_back(&l); }
  cp_circuit circ[3];
  for (int i = 0; i < 3; i++) {
    circ[i].ctx = &ctx;
    memset(&circ[i].sh, 0, sizeof(cp_shape));
    circ[i].sh.degree_bits = i == 2 ? 13 : 12;  // circuits 0 and 1 share a shape, circuit 2 has its own
    circ[i].gates = {{1, 2, 3}};
    circ[i].id = i;
  }
  cp_batcher *b = cp_batcher_create(&ctx, (size_t)max_batch, (unsigned)linger);
  if (!b) { fprintf(stderr, "create: %s\n", cp_last_error(nullptr)); return 1; }
  std::atomic<long> wrong{0}, failed_as_expected{0};
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++)
    th.emplace_back([&, t] {
      for (int k = 0; k < calls; k++) {
        const bool bad = (t * 131 + k * 17) % 97 == 0;
        uint64_t wire = bad ? 0xBAD : ((uint64_t)t << 32 | (uint64_t)k) + 1, pi = 0;
        cp_circuit *c = &circ[(t + k) % 3];
        uint8_t *out = nullptr;
        size_t len = 0;
        const int rc = cp_batcher_prove(b, c, &wire, &pi, 1, 0, 0, &out, &len);
        if (bad) {
          if (rc == CP_ERR_INVALID_ARG && strstr(cp_last_error(nullptr), "0xBAD") && !out) failed_as_expected++;
          else wrong++;
          continue;
        }
        uint64_t got[2] = {~0ull, ~0ull};
        if (rc == CP_OK && len == 16) memcpy(got, out, 16);
        if (got[0] != (uint64_t)c->id || got[1] != wire) wrong++;
        free(out);
      }
    });
  for (auto &x : th) x.join();
  cp_batcher_stats st;
  cp_batcher_get_stats(b, &st);
  cp_batcher_destroy(b);
  printf("{\"calls\": %llu, \"batches\": %llu, \"proofs\": %llu, \"largest_batch\": %llu, \"retried_singly\": %llu, \"wrong\": %ld, "
         "\"failed_as_expected\": %ld, \"violations\": %d}\n",
         (unsigned long long)st.calls, (unsigned long long)st.batches, (unsigned long long)st.proofs, (unsigned long long)st.largest_batch,
         (unsigned long long)st.retried_singly, wrong.load(), failed_as_expected.load(), g_violations.load());
  return wrong.load() || g_violations.load() ? 1 : 0;
}
