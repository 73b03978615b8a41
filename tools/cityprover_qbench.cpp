// cityprover-qbench: native measurement harness above the C ABI (SURVEY.md §8(b) "the build's own C++ harness for
// measurement without Rust"). It is to libcityprover_hip.so what city-rollup's worker loop is to plonky2:
//   * one host thread + one cp_ctx per worker, the circuits loaded once per context and kept resident
//     (CRWorkerToolboxRootCircuits::new, city_rollup_circuit/src/worker/toolbox/root.rs:75-139);
//   * a shared ready queue from which a worker pops jobs, proves them and releases their dependents
//     (SimpleActorWorker::process_next_job, city_rollup_core_worker/src/actors/simple.rs:32-106; the q-bench loop
//     city_rollup_core_worker/src/qbench.rs:44-61) - here up to --batch ready proofs become ONE cp_prove_batch_host call.
// Input: the case file written by tools/dump_qbench_case.py (synthetic qbench-shaped circuits, witnesses, the CPU
// oracle's proof bytes for them, and the example block's 64-proof DAG). Output: one JSON line.
//   --mode throughput : every worker proves --iters batches of --batch independent proofs (common start, wall clock
//                       until the last worker finishes)
//   --mode dag        : --blocks example blocks in flight, proof-level dependencies honoured
//   --lanes L         : cp_ctx_set_lanes(L) on every worker context (internal pipelining of one call; default 1)
// Before timing, each distinct circuit is proved once and the bytes are compared with the oracle's, then cp_verify'd.
// Build: g++ -O2 -std=c++17 -Iinclude tools/cityprover_qbench.cpp -Lcity-rollup_amd -lcityprover_hip
//            -Wl,-rpath,'$ORIGIN/../city-rollup_amd' -lpthread -o tools/cityprover_qbench
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cityprover.h"

namespace {

struct CircuitCase {
  uint64_t digest[4];
  std::vector<uint64_t> cs_values, public_inputs, wires;
  std::vector<uint8_t> expected_proof;
};
struct Case {
  cp_shape shape;
  std::vector<cp_gate> gates;
  int num_selectors = 0;
  std::vector<CircuitCase> circuits;
  std::vector<std::vector<uint32_t>> dag;  // task -> tasks it waits for
};

[[noreturn]] void die(const std::string &msg) {
  fprintf(stderr, "cityprover-qbench: %s\n", msg.c_str());
  exit(1);
}
void rd(FILE *f, void *dst, size_t bytes) {
  if (bytes && fread(dst, 1, bytes, f) != bytes) die("case file truncated");
}
template <class T> T rd(FILE *f) { T v; rd(f, &v, sizeof v); return v; }

Case load_case(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) die(std::string("cannot open ") + path);
  char magic[8];
  rd(f, magic, 8);
  if (memcmp(magic, "CPQBENCH", 8) != 0 || rd<uint32_t>(f) != 1) die("not a version-1 case file");
  Case c;
  int32_t sh[22];
  rd(f, sh, sizeof sh);
  memset(&c.shape, 0, sizeof c.shape);
  c.shape.degree_bits = sh[0]; c.shape.num_constants = sh[1]; c.shape.num_routed_wires = sh[2]; c.shape.num_wires = sh[3];
  c.shape.num_challenges = sh[4]; c.shape.num_partial_products = sh[5]; c.shape.quotient_degree_factor = sh[6];
  c.shape.rate_bits = sh[7]; c.shape.cap_height = sh[8]; c.shape.pow_bits = sh[9]; c.shape.num_query_rounds = sh[10];
  c.shape.n_arity = sh[11];
  for (int i = 0; i < 8; i++) c.shape.arity_bits[i] = sh[12 + i];
  c.shape.zero_knowledge = sh[20];
  c.shape.num_public_inputs = sh[21];
  const uint32_t n_gates = rd<uint32_t>(f);
  c.num_selectors = (int)rd<uint32_t>(f);
  c.gates.resize(n_gates);
  for (auto &g : c.gates) {
    int32_t v[7];
    rd(f, v, sizeof v);
    g.type = v[0]; g.selector_index = v[1]; g.group_start = v[2]; g.group_end = v[3]; g.param = v[4]; g.param2 = v[5]; g.param3 = v[6];
  }
  c.circuits.resize(rd<uint32_t>(f));
  for (auto &k : c.circuits) {
    rd(f, k.digest, sizeof k.digest);
    uint64_t rows = rd<uint64_t>(f), cols = rd<uint64_t>(f);
    k.cs_values.resize(rows * cols);
    rd(f, k.cs_values.data(), k.cs_values.size() * 8);
    k.public_inputs.resize(rd<uint32_t>(f));
    rd(f, k.public_inputs.data(), k.public_inputs.size() * 8);
    rows = rd<uint64_t>(f); cols = rd<uint64_t>(f);
    k.wires.resize(rows * cols);
    rd(f, k.wires.data(), k.wires.size() * 8);
    k.expected_proof.resize(rd<uint64_t>(f));
    rd(f, k.expected_proof.data(), k.expected_proof.size());
  }
  c.dag.resize(rd<uint32_t>(f));
  for (auto &deps : c.dag) {
    deps.resize(rd<uint32_t>(f));
    rd(f, deps.data(), deps.size() * 4);
  }
  fclose(f);
  return c;
}

// one worker = one context with its own resident circuits and page-locked copies of the witnesses
struct Worker {
  cp_ctx *ctx = nullptr;
  std::vector<cp_circuit *> circuits;
  std::vector<uint64_t *> wires;  // page-locked (cp_host_alloc): DMA copies that overlap other contexts
  const Case *cs = nullptr;

  void check(int rc, const char *what) const {
    if (rc != CP_OK) die(std::string(what) + ": " + cp_last_error(ctx));
  }
  void open(const Case &c, int device) {
    cs = &c;
    ctx = cp_ctx_create(device);
    if (!ctx) die(std::string("cp_ctx_create: ") + cp_last_error(nullptr));
    for (const auto &k : c.circuits) {
      cp_circuit *circ = cp_circuit_load(ctx, &c.shape, k.digest, k.cs_values.data(), nullptr);
      if (!circ) die(std::string("cp_circuit_load: ") + cp_last_error(ctx));
      check(cp_circuit_set_gates(circ, c.gates.data(), c.gates.size(), c.num_selectors), "cp_circuit_set_gates");
      circuits.push_back(circ);
      void *p = nullptr;
      check(cp_host_alloc(ctx, k.wires.size() * 8, &p), "cp_host_alloc");
      memcpy(p, k.wires.data(), k.wires.size() * 8);
      wires.push_back((uint64_t *)p);
    }
  }
  // proves the circuits `which` as one batch; returns total proof bytes (proofs are freed unless `keep`)
  size_t prove(const std::vector<uint32_t> &which, std::vector<std::vector<uint8_t>> *keep = nullptr) {
    const size_t B = which.size();
    std::vector<cp_circuit *> cc(B);
    std::vector<const uint64_t *> pis(B), ws(B);
    std::vector<size_t> npi(B), lens(B);
    std::vector<uint8_t *> out(B, nullptr);
    for (size_t i = 0; i < B; i++) {
      const uint32_t k = which[i];
      cc[i] = circuits[k];
      pis[i] = cs->circuits[k].public_inputs.data();
      npi[i] = cs->circuits[k].public_inputs.size();
      ws[i] = wires[k];
    }
    check(cp_prove_batch_host(ctx, B, cc.data(), pis.data(), npi.data(), ws.data(), nullptr, nullptr, out.data(), lens.data()),
          "cp_prove_batch_host");
    size_t total = 0;
    for (size_t i = 0; i < B; i++) {
      total += lens[i];
      if (keep) keep->emplace_back(out[i], out[i] + lens[i]);
      cp_free(out[i]);
    }
    return total;
  }
  void close() {
    for (size_t i = 0; i < circuits.size(); i++) {
      cp_host_free(ctx, wires[i]);
      cp_circuit_destroy(circuits[i]);
    }
    cp_ctx_destroy(ctx);
  }
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct StartGate {  // common start for the workers
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0, total = 0;
  bool go = false;
  void arrive() {
    std::unique_lock<std::mutex> l(m);
    if (++waiting == total) { go = true; cv.notify_all(); }
    cv.wait(l, [&] { return go; });
  }
};

// ready-queue scheduler over `blocks` copies of the DAG (group semantics are already expanded to proof level)
struct Dag {
  std::mutex m;
  std::condition_variable cv;
  std::deque<uint32_t> ready;
  std::vector<int> missing;
  std::vector<std::vector<uint32_t>> children;
  size_t done = 0, total = 0;
  Dag(const Case &c, int blocks) {
    const uint32_t per = (uint32_t)c.dag.size();
    total = (size_t)per * blocks;
    missing.assign(total, 0);
    children.assign(total, {});
    for (int b = 0; b < blocks; b++)
      for (uint32_t t = 0; t < per; t++) {
        const uint32_t id = b * per + t;
        missing[id] = (int)c.dag[t].size();
        for (uint32_t d : c.dag[t]) children[b * per + d].push_back(id);
        if (!missing[id]) ready.push_back(id);
      }
  }
  bool take(size_t max_batch, std::vector<uint32_t> &out) {  // false when everything is done
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !ready.empty() || done == total; });
    if (ready.empty()) return false;
    out.clear();
    while (!ready.empty() && out.size() < max_batch) { out.push_back(ready.front()); ready.pop_front(); }
    return true;
  }
  void finish(const std::vector<uint32_t> &ids) {
    std::lock_guard<std::mutex> l(m);
    for (uint32_t id : ids)
      for (uint32_t ch : children[id])
        if (--missing[ch] == 0) ready.push_back(ch);
    done += ids.size();
    cv.notify_all();
  }
};

}  // namespace

int main(int argc, char **argv) {
  std::string path = "tools/qbench_case.bin", mode = "throughput";
  int contexts = 3, batch = 32, iters = 8, blocks = 32, device = 0, lanes = 1;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
    if (a == "--case") path = val();
    else if (a == "--mode") mode = val();
    else if (a == "--contexts") contexts = atoi(val());
    else if (a == "--batch") batch = atoi(val());
    else if (a == "--iters") iters = atoi(val());
    else if (a == "--blocks") blocks = atoi(val());
    else if (a == "--device") device = atoi(val());
    else if (a == "--lanes") lanes = atoi(val());
    else die("unknown argument " + a);
  }
  if (contexts < 1 || batch < 1 || iters < 1 || blocks < 1) die("bad argument value");
  if (cp_device_count() <= 0) die("no HIP device visible: this library has no CPU fallback");
  const Case cs = load_case(path.c_str());
  const uint32_t n_circ = (uint32_t)cs.circuits.size();
  std::vector<Worker> workers(contexts);
  for (auto &w : workers) {
    w.open(cs, device);
    w.check(cp_ctx_set_lanes(w.ctx, lanes), "cp_ctx_set_lanes");  // > 1: one call is pipelined inside the library
  }

  // parity gate: every circuit once, bytes against the oracle's, then the library's verifier
  {
    std::vector<uint32_t> all(n_circ);
    for (uint32_t k = 0; k < n_circ; k++) all[k] = k;
    std::vector<std::vector<uint8_t>> got;
    workers[0].prove(all, &got);
    for (uint32_t k = 0; k < n_circ; k++) {
      if (got[k] != cs.circuits[k].expected_proof) die("proof bytes of circuit " + std::to_string(k) + " differ from the oracle's");
      workers[0].check(cp_verify(workers[0].circuits[k], got[k].data(), got[k].size()), "cp_verify");
    }
  }
  for (auto &w : workers) {  // warm every context (allocations, staging ring)
    std::vector<uint32_t> warm(batch);
    for (int i = 0; i < batch; i++) warm[i] = i % n_circ;
    w.prove(warm);
  }

  StartGate gate;
  gate.total = contexts + 1;
  std::vector<std::thread> threads;
  std::atomic<size_t> proofs{0}, batches{0};
  double t0 = 0, t1 = 0;
  if (mode == "throughput") {
    for (int t = 0; t < contexts; t++)
      threads.emplace_back([&, t] {
        std::vector<uint32_t> which(batch);
        for (int i = 0; i < batch; i++) which[i] = (t + i) % n_circ;
        gate.arrive();
        for (int it = 0; it < iters; it++) { workers[t].prove(which); proofs += batch; batches++; }
      });
    gate.arrive();
    t0 = now();
    for (auto &th : threads) th.join();
    t1 = now();
  } else if (mode == "dag") {
    Dag dag(cs, blocks);
    for (int t = 0; t < contexts; t++)
      threads.emplace_back([&, t] {
        std::vector<uint32_t> ids, which;
        gate.arrive();
        while (dag.take((size_t)batch, ids)) {
          which.resize(ids.size());
          for (size_t i = 0; i < ids.size(); i++) which[i] = ids[i] % n_circ;
          workers[t].prove(which);
          proofs += ids.size();
          batches++;
          dag.finish(ids);
        }
      });
    gate.arrive();
    t0 = now();
    for (auto &th : threads) th.join();
    t1 = now();
    if (proofs != dag.total) die("scheduler finished early");
  } else {
    die("--mode must be throughput or dag");
  }
  const double dt = t1 - t0;
  const size_t per_block = cs.dag.size();
  printf("{\"harness\": \"cityprover-qbench\", \"mode\": \"%s\", \"contexts\": %d, \"lanes_per_context\": %d, \"max_batch\": %d, \"proofs\": %zu, "
         "\"wall_s\": %.6f, \"proofs_per_s\": %.2f, \"blocks_per_s\": %.3f, \"mean_batch\": %.2f, \"proofs_per_block\": %zu, "
         "\"blocks_in_flight\": %d, \"proof_bytes\": %zu, \"parity\": \"proof bytes == oracle bytes for all %u circuits; cp_verify ok\", "
         "\"wires\": \"host (page-locked), PCIe-inclusive\"}\n",
         mode.c_str(), contexts, lanes, batch, (size_t)proofs, dt, proofs / dt, proofs / dt / (double)per_block,
         (double)proofs / (double)batches, per_block, mode == "dag" ? blocks : 0, cs.circuits[0].expected_proof.size(), n_circ);
  for (auto &w : workers) w.close();
  return 0;
}
