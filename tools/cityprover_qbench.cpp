// cityprover-qbench: the q-bench harness of city-rollup on the MI355X prover, plain C++ above the C ABI
// (include/cityprover.h). It is to libcityprover_hip.so what the reference's offline benchmark is to plonky2:
//
//   city-rollup-cli q-bench -i <dump...> -o <out.json> [-n N]          city_common/src/cli/args.rs:104-117
//   run_qbench / run_worker_qbench                                      city_rollup_core_worker_qbench/src/qbench.rs:15-85
//   SimpleActorWorker::process_next_job / process_job                   city_rollup_core_worker/src/actors/simple.rs:32-115
//   CityEventProcessorMemory (VecDeque job queue, benchmarks)           city_rollup_common/src/actors/simple/events.rs:8-56
//
// What it does, like the reference: deserialise each `BlockProofStoreDump` (bincode), per iteration re-plan the block's job
// DAG into the proof store (plan_jobs), enqueue the leaf jobs, and drain the queue — a job is proved, its output stored
// under `get_output_id()`, its duration recorded, its group counter incremented and, at the goal, its next jobs enqueued.
// Output (-o): `[{"job_id": <24 bytes hex>, "duration": <ms>}]`, serde_json's pretty form, in completion order.
//
// What differs, by design:
//   * the queue is drained by a POOL of workers — one host thread + one cp_ctx per worker, --contexts workers on each of
//     --devices (default: every visible GPU) — pulling one shared ready queue (the in-process form of N worker processes
//     on one Redis queue, city_rollup_core_worker/src/lib.rs:131-145); --contexts 1 --devices 0 --batch 1 is the
//     reference's single-threaded loop, job for job in the same order;
//   * a worker takes up to --batch ready STAGES of jobs whose circuits may share a launch (cp_circuits_batch_compatible: one
//     shape, one gate set — whatever their circuit TYPES) and proves them in one cp_prove_batch_host call; a job of k proofs
//     (proofs_per_job) goes through the queue k times, its next stage ahead of what has not started (a job's `duration` runs
//     from the start of its first stage to the end of its last). The ready stage with the longest chain of dependent proofs
//     behind it is served first; a short queue is shared among the workers that hold no work (Scheduler::take);
//   * --blocks-in-flight F replays F (dump, iteration) instances concurrently, each with its own proof store
//     (BASELINE.json configs[3]: independent blocks), in waves of F (--sliding: a window of F, a block that completes
//     starts the next — measured slower: blocks out of step leave small launches); F = 1 is the reference's
//     one-block-at-a-time loop;
//   * circuits come from a circuit pack (tools/qbench/pack.h): `CircuitData` cannot be built without Rust, so every job
//     type is bound to circuit files + witnesses — dumped from the real worker, or the synthetic shape-equivalent ones —
//     and witness generation is not part of what is timed. The DATA dependencies are real all the same: a
//     job needs its witness and the output proofs it names (tools/qbench/jobs.h proof_dependencies) in the store, or fails.
//   * the Groth16 job (WrapFinalSigHashProofBLS12381) proves its plonky2 wrapper stage and stores the all-zero
//     CityGroth16ProofData of the reference's GROTH16_DISABLED_DEV_MODE (toolbox/root.rs:287-294): no gnark circuit or
//     proving key exists in the tree. --groth16-log-size L adds the Groth16 prover kernels on a synthetic key of 2^L
//     constraints (five MSMs + quotient + the 192-byte packing): the cost of that stage, not a valid proof.
//   * the SHA-256 STARK a sighash job proves before its first plonky2 proof (sighash.rs:132-146) is left out unless
//     --stark-log-rows K is given: then cp_stark_prove runs on a synthetic AIR of the reference's shape (qbench/stark_stage.h),
//     three times per block — the cost of that stage, not a valid proof either (the AIR lives in an absent crate).
//   * the reference re-plans every iteration but never resets `counters` (memory_proof_store/mod.rs:77-83), so from the
//     second iteration on no group ever reaches its goal again and only leaf jobs run; here every iteration starts from
//     fresh counters (--ref-counters keeps the reference's behaviour).
//
// Checks: every job of a block proves its OWN witness (the pack binds one witness per job: SURVEY.md section 8(d) M1). Before
// the clock starts every distinct (circuit, witness) pair is proved once: a proof whose witness file records bytes (the CPU
// oracle's, or the Rust prover's under nonce injection) must equal them, every other one must pass cp_verify; in the timed
// run every proof is compared byte for byte with the bytes that passed. --dry-run runs the whole schedule without proving anything and
// without a GPU (tests of the planner and the queue semantics); --mode throughput is the raw proofs/s measurement;
// --mode callers measures one-job-per-call threads (--callers T) merged by a cp_batcher (--batch = its max_batch, --linger-us);
// --callers T in the default mode drains the DAG with T such threads per context instead of one batching thread;
// --mode redis-worker --redis HOST:PORT [--drain] [--max-jobs N] is one `l2-worker` of a live deployment: jobs from the RSMQ
// "JOB" queue, witnesses / proofs / counters in the Redis proof store (tools/qbench/redis.h), one job at a time as the
// reference's worker loop does (city_rollup_core_worker/src/lib.rs:104-146).
// Build: g++ -O2 -std=c++17 -Iinclude tools/cityprover_qbench.cpp -Lcity-rollup_amd -lcityprover_hip
//            -Wl,-rpath,'$ORIGIN/../city-rollup_amd' -lpthread -o tools/cityprover_qbench
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cityprover.h"
#include "qbench/jobs.h"
#include "qbench/stark_stage.h"
#include "qbench/pack.h"
#include "qbench/redis.h"

namespace {

using qb::JobId;

[[noreturn]] void die(const std::string &msg) {
  fprintf(stderr, "cityprover-qbench: %s\n", msg.c_str());
  exit(1);
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Options {
  std::vector<std::string> inputs;
  std::string output, network = "dogeregtest", pack_dir, mode = "qbench", trace_path;
  std::string redis_uri;  // --mode redis-worker: the deployment's Redis (proof store + RSMQ queues)
  bool drain = false;     // redis-worker: leave when the JOB queue is empty instead of polling for ever
  int max_jobs = 0;       // redis-worker: leave after this many jobs (0: no limit)
  int redis_batch = 1;    // redis-worker: messages taken per round (1 = the reference's loop)
  int iterations = 1, contexts = 3, batch = 128, blocks_in_flight = 1, lanes = 1, iters = 8, callers = 0, linger_us = 0;
  bool dry_stages = false;  // --dry-run-stages: every stage of a job is a queue entry, as in a real run
  int dry_job_us = 0;   // --dry-run only: pretend a proving batch takes this long, so that the queue is shared among the worker slots
  int groth16_log = 0;  // > 0: the Groth16 job runs cp_groth16_prove_bls12381 on a synthetic key of 2^groth16_log constraints
  int stark_log_rows = 0;  // > 0: every GenerateSigHashIntrospectionProof job first proves a STARK of 2^stark_log_rows rows (qbench/stark_stage.h)
  int stark_contexts = 0;  // > 0: that many EXTRA contexts per device prove nothing but STARK stages, and the --contexts workers none of them
  std::vector<int> devices;  // empty: all visible
  bool dry_run = false, ref_counters = false, check_plan = false;
  bool sliding = false;    // --sliding: a window of --blocks-in-flight blocks (a block that completes starts the next) instead of waves
  bool skip_gate = false;  // --mode throughput only: no check of the proofs (counter-collection runs: every launch then carries a full batch)
};

// ---- one (dump, iteration) replay ---------------------------------------------------------------------------------
struct Instance {
  size_t index = 0, dump_index = 0;
  int iteration = 0;
  qb::ProofStore store;
  std::mutex m;  // guards `store`
  bool complete = false;
  size_t jobs_done = 0, proofs_done = 0;
  double t_start = 0, t_end = 0;
  std::unordered_map<JobId, int, qb::JobIdHash> tail_memo;  // counter id -> proofs on the longest chain after a job of that group

  // Proofs on the longest dependency chain from `job` to the end of the block, `job` included: what the block's latency is
  // made of when few jobs are ready. Read off the plan's own records (the next-jobs list of the job's group). Call with `m` held.
  int chain_length(const JobId &job) {
    if (job.topic == qb::NotifyOrchestratorComplete) return 0;
    const int own = job.topic == qb::GenerateStandardProof ? qb::proofs_per_job(job.circuit_type) : 0;
    const JobId key = job.counter_id();
    auto it = tail_memo.find(key);
    if (it == tail_memo.end()) {
      int best = 0;
      tail_memo[key] = 0;  // a cycle in a malformed plan ends here
      if (store.has(job.next_jobs_id_of_counter()))
        for (const JobId &n : store.get_next_jobs(job)) best = std::max(best, chain_length(n));
      it = tail_memo.find(key);
      it->second = best;
    }
    return own + it->second;
  }
};

// one STAGE of a job: a job of k proofs (proofs_per_job) goes through the queue k times, stage s + 1 after stage s is proved, so
// that a worker is never held by the later stages of a long job and every launch can take whatever stages are ready
struct QueueEntry { Instance *inst; JobId job; int chain = 0; int stage = 0; double t_first = 0; };
struct BenchRecord { JobId job; uint64_t duration_ms; double t0, t1; int worker, batch; size_t instance; };

// the shared ready queue (CityEventProcessorMemory::job_queue) + the bookkeeping of the run
struct Scheduler {
  std::mutex m;
  std::condition_variable cv;
  std::deque<QueueEntry> queue;
  size_t in_flight = 0;        // jobs taken and not yet finished
  size_t pending_instances = 0;  // started and not complete
  bool failed = false;
  std::string error;
  std::vector<BenchRecord> benchmarks;  // completion order
  std::vector<JobId> processed;        // every job popped, in pop order (barrier and notify jobs included)
  int stage_class[256][8] = {{0}};     // (circuit type, stage) -> batch-compatibility class of that stage's circuit
  int cls(const QueueEntry &e) const { return e.stage < 0 ? -2 : stage_class[e.job.circuit_type][e.stage < 8 ? e.stage : 7]; }
  size_t dry_stark_stages = 0;         // --dry-run --stark-log-rows: STARK stages that went through the queue
  bool stark_stage = false;            // --stark-log-rows: a sighash job enters the queue at stage -1 = its STARK (a unit of its own: the
                                       // three of a block go to three workers instead of one after the other on whoever took the jobs)
  size_t n_workers = 1;                // threads draining the queue
  size_t busy = 0;                     // of them, holding work (between the take() that returned it and their next take())
  std::function<void()> on_block_complete;  // --sliding: starts the next block

  void enqueue(Instance *inst, const std::vector<JobId> &jobs) {
    std::vector<int> chain(jobs.size());
    {
      std::lock_guard<std::mutex> li(inst->m);
      for (size_t i = 0; i < jobs.size(); i++) chain[i] = inst->chain_length(jobs[i]);
    }
    std::lock_guard<std::mutex> l(m);
    for (size_t i = 0; i < jobs.size(); i++) {
      const JobId &j = jobs[i];
      const bool stark_first = stark_stage && j.topic == qb::GenerateStandardProof && j.circuit_type == qb::GenerateSigHashIntrospectionProof;
      queue.push_back({inst, j, chain[i] + (stark_first ? 1 : 0), stark_first ? -1 : 0});
      if (roles) cv.notify_all();
      else cv.notify_one();  // one sleeper per job: with a hundred worker threads, waking them all for every job is what costs
    }
  }
  // the next stage of a job whose stage has just been proved: ahead of what has not started yet
  void requeue(const QueueEntry &e) {
    std::lock_guard<std::mutex> l(m);
    queue.push_front(e);
    if (roles) cv.notify_all();
    else cv.notify_one();
  }
  void fail(const std::string &msg) {
    std::lock_guard<std::mutex> l(m);
    if (!failed) { failed = true; error = msg; }
    cv.notify_all();
  }
  // Blocks until work is ready. Takes the front job and, when it is a proving job, up to max_batch - 1 further ready jobs
  // whose circuits may share a launch with it (FIFO among them). Returns false when the run is over (all instances complete,
  // or a failure).
  bool roles = false;  // --stark-contexts: workers take by role, so every enqueue wakes all sleepers (the one woken may not be the one who can take it)
  static bool fits(const QueueEntry &e, int role) { return role == 0 || (role == 2) == (e.stage < 0); }
  bool take(size_t max_batch, std::vector<QueueEntry> &out, int role = 0) {
    std::unique_lock<std::mutex> l(m);
    static thread_local bool holds_work = false;  // a worker is busy from the take() that gave it work to its next take()
    if (holds_work) { if (role != 2) busy--; holds_work = false; }
    for (;;) {
      if (failed) return false;
      bool any = false;
      for (const auto &e : queue) if (fits(e, role)) { any = true; break; }
      if (any) break;
      if (in_flight == 0 && pending_instances == 0) return false;
      if (in_flight == 0 && queue.empty() && pending_instances > 0) {
        // nothing running, nothing ready, blocks unfinished: the DAG cannot make progress
        failed = true;
        error = "the job queue ran dry before every block completed (a group never reached its goal)";
        cv.notify_all();
        return false;
      }
      cv.wait(l);
    }
    out.clear();
    static const bool share_short_queues = !getenv("CITYPROVER_QBENCH_NO_SHARE");
    static const bool longest_chain_first = !getenv("CITYPROVER_QBENCH_FIFO");
    if (longest_chain_first && max_batch > 1 && n_workers > 1)
      // Order of service: the ready job with the longest chain of dependent proofs behind it first — the five-stage
      // introspection jobs the planner enqueues ahead of the leaves have a shorter way to the end of the block than the leaves.
      // One block alone: 71 -> 67 ms; with many blocks in flight never slower than first-in-first-out
      // (profiles/r03_qbench_chain_ab.jsonl). NOT "the older block first": blocks that advance in step fill the launches
      // (mean launch 75 proofs at 64 blocks in flight); served oldest-first they drift apart, ready jobs trickle in and the
      // launches shrink to 9 (34 -> 28 blocks/s, profiles/r03_qbench_window_ab_oldest_first.jsonl).
      // CITYPROVER_QBENCH_FIFO=1 restores the queue order.
      std::stable_sort(queue.begin(), queue.end(), [](const QueueEntry &a, const QueueEntry &b) { return a.chain > b.chain; });
    {
      auto it = queue.begin();
      while (!fits(*it, role)) ++it;
      out.push_back(*it);
      queue.erase(it);
    }
    const JobId first = out[0].job;
    if (out[0].stage < 0) max_batch = 1;  // a STARK is proved alone
    if (first.topic == qb::GenerateStandardProof && max_batch > 1 && n_workers > 1 && share_short_queues) {
      // A queue that the free workers could empty between them is SHARED among them instead of going to whoever woke first: one
      // block alone, twenty-three ready stages, is three launches on three contexts at once, not one launch while two contexts
      // idle (67 ms against 80 ms per block); with 4 blocks in flight +14 %, with 8 +2.5 %, from 16 up the queue is long enough
      // for full launches and nothing changes (profiles/r03_qbench_share_ab.jsonl; with round 2's cap of 32 proofs per launch
      // the rule cost 6-8 % there and was limited to runs with fewer blocks than workers).
      size_t ready = 1;
      for (const auto &e : queue)
        if (e.job.topic == qb::GenerateStandardProof && cls(e) == cls(out[0])) ready++;
      // ... among the workers that hold no work right now (this one included): the first of three takes a third, the second half
      // of what is left, the third the rest — with ready / n_workers for everyone the shares shrank with the queue (23 ready
      // leaves went out as 8 + 5 + 4 and six waited for the next free worker: profiles/r03_one_block_timeline.txt)
      const size_t takers = n_workers > busy ? n_workers - busy : 1;
      const size_t share = (ready + takers - 1) / takers;
      if (share < max_batch) max_batch = share < 1 ? 1 : share;
    }
    if (first.topic == qb::GenerateStandardProof)
      for (auto it = queue.begin(); it != queue.end() && out.size() < max_batch;) {
        if (it->job.topic == qb::GenerateStandardProof && cls(*it) == cls(out[0]) && fits(*it, role)) {
          out.push_back(*it);
          it = queue.erase(it);
        } else {
          ++it;
        }
      }
    for (const auto &e : out)
      if (e.stage == 0) processed.push_back(e.job);
    in_flight += out.size();
    if (role != 2) busy++;   // `busy` counts the workers that share plonky2 launches
    holds_work = true;
    return true;
  }
  void finished(size_t n) {
    std::lock_guard<std::mutex> l(m);
    in_flight -= n;
    if (in_flight == 0) cv.notify_all();  // the end of the run (or a DAG that ran dry) is decided by the sleepers
  }
};

// ---- the Groth16 stage of the WrapFinalSigHashProofBLS12381 job, on a SYNTHETIC proving key ----------------------------
// The reference proves a gnark circuit that verifies the wrapper's plonky2 proof (toolbox/root.rs:296-304); neither that
// circuit nor its proving key exists in the tree, so with --groth16-log-size L the harness measures the
// prover kernels on a stand-in of the same SHAPE: point sets (a i + b) G built on the device, a witness with 60 % of its
// wires in {0, 1} (what R1CS witnesses look like), random evaluation vectors. The output goes through the real tail:
// cp_groth16_proof_pack_city -> CityGroth16ProofData -> bincode (four hex strings), stored like the reference stores it.
// The proof is NOT a valid proof of anything (no circuit) — correctness of the assembly is tests/test_gpu_groth16.py.
static const uint64_t BLS_G1[12] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull,
                                    0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull, 0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull,
                                    0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
static const uint64_t BLS_G2[24] = {0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull,
    0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull,
    0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull,
    0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull};

struct Groth16Stage {
  int log_n = 0;
  cp_groth16_pk pk;
  void *sets[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  void *witness = nullptr, *ev_master = nullptr, *ev[3] = {nullptr, nullptr, nullptr};

  void open(cp_ctx *ctx, int log_size) {
    log_n = log_size;
    const size_t n = (size_t)1 << log_n;
    auto check = [&](int rc, const char *what) { if (rc != CP_OK) throw std::runtime_error(std::string(what) + ": " + cp_last_error(ctx)); };
    const uint32_t ab[5][2] = {{3, 1}, {5, 2}, {7, 3}, {11, 4}, {13, 5}};
    for (int k = 0; k < 5; k++) {
      const bool g2 = k == 2;
      check(cp_dev_alloc(ctx, n * (g2 ? CP_G2_AFFINE_BYTES : CP_G1_AFFINE_BYTES), &sets[k]), "cp_dev_alloc");
      check(g2 ? cp_msm_bls12381_g2_synthetic_points_dev(ctx, BLS_G2, ab[k][0], ab[k][1], n, sets[k])
               : cp_msm_bls12381_g1_synthetic_points_dev(ctx, BLS_G1, ab[k][0], ab[k][1], n, sets[k]), "synthetic points");
    }
    std::vector<uint64_t> w(4 * n), e(4 * n);
    uint64_t x = 0x9E3779B97F4A7C15ull + (uint64_t)log_n;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < n; i++) {
      const bool small = next() % 10 < 6;
      for (int j = 0; j < 4; j++) {
        w[4 * i + j] = small ? (j == 0 ? next() & 1 : 0) : next() & (j == 3 ? ((1ull << 62) - 1) : ~0ull);
        e[4 * i + j] = next() & (j == 3 ? ((1ull << 62) - 1) : ~0ull);
      }
    }
    check(cp_dev_alloc(ctx, n * 32, &witness), "cp_dev_alloc");
    check(cp_h2d(ctx, witness, w.data(), n * 32), "cp_h2d");
    check(cp_dev_alloc(ctx, n * 32, &ev_master), "cp_dev_alloc");
    check(cp_h2d(ctx, ev_master, e.data(), n * 32), "cp_h2d");
    for (auto &p : ev) check(cp_dev_alloc(ctx, n * 32, &p), "cp_dev_alloc");
    memset(&pk, 0, sizeof pk);
    pk.n_wires = n;
    pk.n_private = n - 16;
    pk.log_domain = log_n;
    pk.a_g1 = sets[0]; pk.b_g1 = sets[1]; pk.b_g2 = sets[2]; pk.k_g1 = sets[3]; pk.z_g1 = sets[4];
    memcpy(pk.alpha_g1, BLS_G1, sizeof BLS_G1);
    memcpy(pk.beta_g1, BLS_G1, sizeof BLS_G1);
    memcpy(pk.delta_g1, BLS_G1, sizeof BLS_G1);
    memcpy(pk.beta_g2, BLS_G2, sizeof BLS_G2);
    memcpy(pk.delta_g2, BLS_G2, sizeof BLS_G2);
  }
  // one Groth16 proof -> bincode of CityGroth16ProofData
  std::vector<uint8_t> prove(cp_ctx *ctx, uint64_t seed) {
    auto check = [&](int rc, const char *what) { if (rc != CP_OK) throw std::runtime_error(std::string(what) + ": " + cp_last_error(ctx)); };
    const size_t n = (size_t)1 << log_n;
    for (auto &p : ev) check(cp_d2d(ctx, p, ev_master, n * 32), "cp_d2d");   // the prover overwrites its evaluation vectors
    const uint64_t r[4] = {seed * 2 + 1, 11, 22, 33}, s[4] = {seed * 2 + 2, 44, 55, 66};
    uint64_t a[12], b[24], c[12];
    check(cp_groth16_prove_bls12381(ctx, &pk, (const uint64_t *)witness, (uint64_t *)ev[0], (uint64_t *)ev[1], (uint64_t *)ev[2], r, s, a, b, c),
          "cp_groth16_prove_bls12381");
    uint8_t packed[192];
    if (cp_groth16_proof_pack_city(a, b, c, packed) != CP_OK) throw std::runtime_error(std::string("cp_groth16_proof_pack_city: ") + cp_last_error(nullptr));
    std::vector<uint8_t> v;  // bincode: each Serialized2DFeltBLS12381 is a hex STRING (serde_with::hex::Hex): u64 length 96 + 96 characters
    static const char *d = "0123456789abcdef";
    for (int el = 0; el < 4; el++) {
      const uint64_t len = 96;
      v.insert(v.end(), (const uint8_t *)&len, (const uint8_t *)&len + 8);
      for (int i = 0; i < 48; i++) { v.push_back((uint8_t)d[packed[48 * el + i] >> 4]); v.push_back((uint8_t)d[packed[48 * el + i] & 15]); }
    }
    return v;
  }
  void close(cp_ctx *ctx) {
    for (auto *p : sets) if (p) cp_dev_free(ctx, p);
    if (witness) cp_dev_free(ctx, witness);
    if (ev_master) cp_dev_free(ctx, ev_master);
    for (auto *p : ev) if (p) cp_dev_free(ctx, p);
  }
};

// ---- a worker: one context, its own resident circuits, page-locked witnesses ----------------------------------------
struct Worker {
  int index = 0, device = 0;
  cp_ctx *ctx = nullptr;
  const qb::Pack *pack = nullptr;
  std::vector<cp_circuit *> circuits;   // pack circuit index -> resident circuit of this context
  std::vector<uint64_t *> wires;        // pack witness index -> page-locked copy of the wire matrix
  size_t parity_checked = 0, proofs = 0, groth16_proofs = 0, launches = 0;
  Groth16Stage groth16;
  bool has_groth16 = false;
  qb::StarkStage stark;
  bool has_stark = false;
  size_t stark_proofs = 0, stark_bytes = 0;
  cp_batcher *batcher = nullptr;  // --callers: several threads share this worker and prove one job per call through it

  void check(int rc, const char *what) const {
    if (rc != CP_OK) throw std::runtime_error(std::string(what) + ": " + cp_last_error(ctx));
  }
  int role = 0;  // 0: takes any ready stage; 1: plonky2 stages only; 2: STARK stages only (Scheduler::take)
  void open_bare(int dev) {  // a context without resident circuits: a STARK-only worker
    device = dev;
    ctx = cp_ctx_create(dev);
    if (!ctx) throw std::runtime_error(std::string("cp_ctx_create: ") + cp_last_error(nullptr));
  }
  void open(const qb::Pack &p, int dev, int lanes) {
    pack = &p;
    device = dev;
    ctx = cp_ctx_create(dev);
    if (!ctx) throw std::runtime_error(std::string("cp_ctx_create: ") + cp_last_error(nullptr));
    check(cp_ctx_set_lanes(ctx, lanes), "cp_ctx_set_lanes");
    for (const auto &f : p.circuit_files) {
      cp_circuit *c = cp_circuit_load_file(ctx, f.c_str());
      if (!c) throw std::runtime_error("cp_circuit_load_file(" + f + "): " + cp_last_error(ctx));
      circuits.push_back(c);
    }
    for (size_t w = 0; w < p.witnesses.size(); w++) {
      const qb::Witness &wt = *p.witnesses[w];
      cp_shape sh;
      uint64_t dg[4];
      check(cp_circuit_shape(circuits[p.witness_circuit[w]], &sh, dg), "cp_circuit_shape");
      if (memcmp(dg, wt.digest, 32) != 0) throw std::runtime_error("witness " + std::to_string(w) + " was made for another circuit (digest differs)");
      if ((int)wt.num_wires != sh.num_wires || (int)wt.degree_bits != sh.degree_bits || (int)wt.public_inputs.size() != sh.num_public_inputs)
        throw std::runtime_error("witness " + std::to_string(w) + " does not fit its circuit's shape");
      void *pinned = nullptr;
      check(cp_host_alloc(ctx, wt.wires.size() * 8, &pinned), "cp_host_alloc");
      memcpy(pinned, wt.wires.data(), wt.wires.size() * 8);
      wires.push_back((uint64_t *)pinned);
    }
  }
  // one proof to make: a circuit of the pack on one of its witnesses
  struct Item { int circuit, witness; };
  // Proves `items` as ONE batch (they must be batch-compatible) and holds every proof against the bytes that passed the
  // gate for its witness (Shared::expected; empty while the gate itself runs). Returns the proofs.
  std::vector<std::vector<uint8_t>> prove_items(const std::vector<Item> &items, const std::vector<std::vector<uint8_t>> *expected) {
    const size_t count = items.size();
    std::vector<cp_circuit *> cc(count);
    std::vector<const uint64_t *> pis(count), ws(count);
    std::vector<size_t> npi(count), lens(count);
    std::vector<uint8_t *> out(count, nullptr);
    for (size_t i = 0; i < count; i++) {
      const qb::Witness &wt = *pack->witnesses[items[i].witness];
      cc[i] = circuits[items[i].circuit];
      pis[i] = wt.public_inputs.data();
      npi[i] = wt.public_inputs.size();
      ws[i] = wires[items[i].witness];
    }
    if (batcher) {  // the calling thread is one of several sharing this worker: its message is the thread's, not the context's
      for (size_t i = 0; i < count; i++)
        if (cp_batcher_prove(batcher, cc[i], ws[i], pis[i], npi[i], 0, 0, &out[i], &lens[i]) != CP_OK)
          throw std::runtime_error(std::string("cp_batcher_prove: ") + cp_last_error(nullptr));
    } else {
      check(cp_prove_batch_host(ctx, count, cc.data(), pis.data(), npi.data(), ws.data(), nullptr, nullptr, out.data(), lens.data()),
            "cp_prove_batch_host");
    }
    std::vector<std::vector<uint8_t>> res(count);
    int bad = -1;
    size_t checked = 0;
    for (size_t i = 0; i < count; i++) {
      res[i].assign(out[i], out[i] + lens[i]);
      cp_free(out[i]);
      if (expected && !(*expected)[items[i].witness].empty()) {
        if (res[i] != (*expected)[items[i].witness]) bad = (int)i;
        checked++;
      }
    }
    __atomic_fetch_add(&parity_checked, checked, __ATOMIC_RELAXED);  // plain counters, several threads with --callers
    __atomic_fetch_add(&proofs, count, __ATOMIC_RELAXED);
    __atomic_fetch_add(&launches, (size_t)1, __ATOMIC_RELAXED);
    if (bad >= 0)
      throw std::runtime_error("proof bytes differ from the bytes that passed the gate for this witness (circuit " + pack->circuit_files[items[bad].circuit] + ")");
    return res;
  }
  // The gate every distinct proof passes before the clock starts: all witnesses of a binding in one batch; a witness file
  // that records a proof (the CPU oracle's / the Rust prover's) demands those bytes, every other proof must pass cp_verify.
  // `expected` is filled by the first worker and only compared against by the others.
  void gate(std::vector<std::vector<uint8_t>> &expected, size_t *oracle_checked, size_t *verified) {
    for (const auto &kv : pack->by_type)
      for (const auto &b : kv.second) {
        std::vector<Item> items;
        for (int w : b.witnesses) items.push_back({b.circuit, w});
        const auto proofs_ = prove_items(items, nullptr);
        for (size_t i = 0; i < items.size(); i++) {
          const int w = items[i].witness;
          if (!expected[w].empty()) {  // seen before (another worker, or a witness bound twice)
            if (proofs_[i] != expected[w]) throw std::runtime_error("two provers disagree on the proof of witness " + std::to_string(w));
            continue;
          }
          const qb::Witness &wt = *pack->witnesses[w];
          if (!wt.expected_proof.empty()) {
            if (proofs_[i] != wt.expected_proof)
              throw std::runtime_error("proof bytes differ from the bytes recorded in the witness file (circuit " + pack->circuit_files[b.circuit] + ")");
            if (oracle_checked) ++*oracle_checked;
          } else {
            if (cp_verify(circuits[b.circuit], proofs_[i].data(), proofs_[i].size()) != CP_OK)
              throw std::runtime_error(std::string("cp_verify rejects a proof of ") + pack->circuit_files[b.circuit] + ": " + cp_last_error(ctx));
            if (verified) ++*verified;
          }
          expected[w] = proofs_[i];
        }
      }
  }
  // batch-compatibility classes of the pack's circuits (cp_circuits_batch_compatible against one representative per class)
  std::vector<int> circuit_classes() const {
    std::vector<int> cls(circuits.size(), -1), rep;
    for (size_t c = 0; c < circuits.size(); c++) {
      for (size_t k = 0; k < rep.size() && cls[c] < 0; k++)
        if (cp_circuits_batch_compatible(circuits[c], circuits[rep[k]]) == 1) cls[c] = (int)k;
      if (cls[c] < 0) { cls[c] = (int)rep.size(); rep.push_back((int)c); }
    }
    return cls;
  }
  void close() {
    if (batcher) cp_batcher_destroy(batcher);
    batcher = nullptr;
    if (has_groth16) groth16.close(ctx);
    if (has_stark) stark.close(ctx);
    for (auto *w : wires) cp_host_free(ctx, w);
    for (auto *c : circuits) cp_circuit_destroy(c);
    if (ctx) cp_ctx_destroy(ctx);
    ctx = nullptr;
  }
};

// bincode of CityGroth16ProofData with all-zero elements: 4 x (u64 length 96 + 96 hex characters) — each
// Serialized2DFeltBLS12381 serialises as a hex STRING (serde_with::hex::Hex; city_crypto/src/field/serialized_2d_felt_bls12381.rs:9-11)
std::vector<uint8_t> zero_groth16_bincode() {
  std::vector<uint8_t> v;
  for (int e = 0; e < 4; e++) {
    const uint64_t n = 96;
    v.insert(v.end(), (const uint8_t *)&n, (const uint8_t *)&n + 8);
    v.insert(v.end(), 96, (uint8_t)'0');
  }
  return v;
}

// process_job (actors/simple.rs:57-115) for a batch of jobs of one circuit type, or a single non-proving job
// what every worker of a run shares, read-only while it runs
struct Shared {
  const qb::Pack *pack = nullptr;
  std::vector<int> circuit_class;                       // pack circuit -> batch-compatibility class
  std::vector<std::vector<uint8_t>> expected;           // pack witness -> the proof bytes that passed the gate
  std::vector<std::unordered_map<JobId, size_t, qb::JobIdHash>> ordinals;  // per dump: proving job -> its index among the block's jobs of its type
};

void process_batch(const Options &opt, Scheduler &S, Worker *worker, const Shared &shared, const std::vector<QueueEntry> &batch) {
  const JobId first = batch[0].job;
  const qb::Pack *pack = shared.pack;
  const std::vector<int> &circuit_class = shared.circuit_class;
  const std::vector<std::vector<uint8_t>> *expected = &shared.expected;
  auto job_ordinal = [&](const QueueEntry &e) -> size_t {
    const auto &m = shared.ordinals[e.inst->dump_index];
    auto it = m.find(e.job);
    return it == m.end() ? 0 : it->second;
  };
  const double t0 = now_s();
  static const bool whole_jobs = getenv("CITYPROVER_QBENCH_WHOLE_JOBS") != nullptr;  // A/B: a worker keeps a job through all its stages
  auto t_first = [&](size_t i) { return batch[i].t_first > 0 ? batch[i].t_first : t0; };  // when the job's first stage started
  if (batch[0].stage < 0) {  // the STARK of a sighash job (sighash.rs:132-146): proved, then the job goes on with its first plonky2 proof
    for (const auto &e : batch) {
      if (opt.dry_run) {  // the schedule alone: the stage takes the dry run's job time, five times over (a STARK is the longest unit)
        if (opt.dry_job_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(5 * opt.dry_job_us));
        std::lock_guard<std::mutex> l(S.m);
        S.dry_stark_stages++;
      } else {
        if (!worker || !worker->has_stark) throw std::runtime_error("a STARK stage was scheduled without a STARK prover");
        worker->stark_bytes += worker->stark.prove(worker->ctx, (uint64_t)e.job.goal_id * 16 + e.job.task_index);
        worker->stark_proofs++;
      }
      S.requeue({e.inst, e.job, e.chain > 0 ? e.chain - 1 : 0, 0, t0});
    }
    return;
  }
  std::vector<std::vector<uint8_t>> outputs(batch.size());
  std::vector<char> done(batch.size(), 0);
  // the tail of process_job (actors/simple.rs:89-106) for one job of the batch: store the output, record the duration, count,
  // release what waits for it. Called as soon as the job's LAST stage is proved — a one-proof job does not wait for the five
  // stages of a sighash job that happened to share its launch.
  auto finish = [&](size_t i) {
    if (done[i]) return;
    done[i] = 1;
    Instance *inst = batch[i].inst;
    const JobId job = batch[i].job;
    if (job.topic == qb::GenerateStandardProof && job.circuit_type == qb::WrapFinalSigHashProofBLS12381) {
      if (worker && worker->has_groth16) {  // the Groth16 prover kernels on a synthetic key (Groth16Stage)
        outputs[i] = worker->groth16.prove(worker->ctx, job.goal_id * 16 + job.task_index);
        worker->groth16_proofs++;
      } else {
        outputs[i] = zero_groth16_bincode();  // GROTH16_DISABLED_DEV_MODE (toolbox/root.rs:287-294)
      }
    }
    const double t1 = now_s();
    const uint64_t ms = (uint64_t)((t1 - t_first(i)) * 1e3);
    std::vector<JobId> release;
    {
      std::lock_guard<std::mutex> l(inst->m);
      if (job.topic == qb::GenerateStandardProof) {
        inst->store.set_bytes(job.output_id(), outputs[i]);
        inst->jobs_done++;
        inst->proofs_done += (size_t)qb::proofs_per_job(job.circuit_type);
      }
      if (job.topic == qb::NotifyOrchestratorComplete) {
        inst->complete = true;
        inst->t_end = t1;
      } else {
        const uint32_t goal = inst->store.get_goal(job);
        if (goal != 0 && inst->store.inc_counter(job.counter_id()) == goal) release = inst->store.get_next_jobs(job);
      }
    }
    {
      std::lock_guard<std::mutex> l(S.m);
      if (job.topic == qb::GenerateStandardProof)
        S.benchmarks.push_back({job, ms, t_first(i), t1, worker ? worker->index : -1, (int)batch.size(), inst->index});
      if (job.topic == qb::NotifyOrchestratorComplete) S.pending_instances--;
    }
    if (job.topic == qb::NotifyOrchestratorComplete && S.on_block_complete) S.on_block_complete();  // this job still counts as in flight
    if (!release.empty()) S.enqueue(inst, release);
  };
  if (first.topic == qb::GenerateStandardProof) {
    // inputs: the job's witness and every proof it names must be in the store (worker/traits.rs:74-83,164-202)
    for (const auto &e : batch) {
      if (e.stage != 0) continue;  // checked when the job's first stage was taken
      std::lock_guard<std::mutex> l(e.inst->m);
      const std::vector<uint8_t> &w = e.inst->store.get_bytes(e.job);
      for (const JobId &dep : qb::proof_dependencies(e.job, w))
        if (e.inst->store.get_bytes(dep).empty()) throw qb::StoreError("Proof " + dep.hex() + " needed by " + e.job.hex() + " is empty");
    }
    if (!opt.dry_run) {
      // the stages of a job are a chain (stage s + 1 verifies the proof of stage s); the jobs of a batch may be of different
      // types with different numbers of stages: stage s is proved for every job that has one, one launch per
      // batch-compatibility class of the stage's circuit
      int max_stages = 0;
      std::vector<const std::vector<qb::Binding> *> st(batch.size());
      for (size_t i = 0; i < batch.size(); i++) {
        const uint8_t ct = batch[i].job.circuit_type;
        st[i] = &pack->stages_for(ct);
        if ((int)st[i]->size() != qb::proofs_per_job(ct))
          throw std::runtime_error("the pack binds " + std::to_string(st[i]->size()) + " stages to circuit type " + std::to_string(ct) +
                                   ", the job proves " + std::to_string(qb::proofs_per_job(ct)));
        max_stages = std::max(max_stages, (int)st[i]->size());
      }
      if (!whole_jobs) {
        // ONE stage of every job of the batch (take() put stages of one compatibility class together): a job whose last stage
        // this was is finished, the others go back to the queue with their next stage
        std::vector<Worker::Item> items;
        for (size_t i = 0; i < batch.size(); i++) {
          if (batch[i].stage >= (int)st[i]->size()) throw std::runtime_error("stage out of range for " + batch[i].job.hex());
          const qb::Binding &b = (*st[i])[(size_t)batch[i].stage];
          items.push_back({b.circuit, b.witness_for(job_ordinal(batch[i]))});
        }
        auto proofs = worker->prove_items(items, expected);
        for (size_t i = 0; i < batch.size(); i++) {
          if (batch[i].stage + 1 == (int)st[i]->size()) {
            outputs[i] = std::move(proofs[i]);
            finish(i);
          } else {
            done[i] = 1;  // not finished: its next stage is a queue entry of its own
            S.requeue({batch[i].inst, batch[i].job, batch[i].chain > 0 ? batch[i].chain - 1 : 0, batch[i].stage + 1, t_first(i)});
          }
        }
        max_stages = 0;
      }
      for (int s = 0; s < max_stages; s++) {
        std::vector<std::pair<int, size_t>> order;  // (class of the stage's circuit, position in the batch)
        for (size_t i = 0; i < batch.size(); i++)
          if (s < (int)st[i]->size()) order.push_back({circuit_class[(*st[i])[s].circuit], i});
        std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (size_t lo = 0; lo < order.size();) {
          size_t hi = lo;
          std::vector<Worker::Item> items;
          while (hi < order.size() && order[hi].first == order[lo].first) {
            const size_t i = order[hi].second;
            const qb::Binding &b = (*st[i])[s];
            items.push_back({b.circuit, b.witness_for(job_ordinal(batch[i]))});
            hi++;
          }
          auto proofs = worker->prove_items(items, expected);
          for (size_t k = lo; k < hi; k++)
            if (s + 1 == (int)st[order[k].second]->size()) {
              outputs[order[k].second] = std::move(proofs[k - lo]);
              finish(order[k].second);
            }
          lo = hi;
        }
      }
    } else {
      for (auto &o : outputs) o.assign(1, 0);  // placeholder: "an output exists"
      if (opt.dry_job_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(opt.dry_job_us));
      if (opt.dry_stages)  // the stage machinery without a GPU: a job of k proofs goes through the queue k times
        for (size_t i = 0; i < batch.size(); i++)
          if (batch[i].stage + 1 < qb::proofs_per_job(batch[i].job.circuit_type)) {
            done[i] = 1;
            S.requeue({batch[i].inst, batch[i].job, batch[i].chain > 0 ? batch[i].chain - 1 : 0, batch[i].stage + 1, t_first(i)});
          }
    }
  }
  for (size_t i = 0; i < batch.size(); i++) finish(i);
}

void worker_loop(const Options &opt, Scheduler &S, Worker *worker, const Shared &shared, size_t take, std::atomic<size_t> *jobs_of_slot, int role) {
  std::vector<QueueEntry> batch;
  while (S.take(take, batch, role)) {
    try {
      process_batch(opt, S, worker, shared, batch);
      if (jobs_of_slot && batch[0].job.topic == qb::GenerateStandardProof) jobs_of_slot->fetch_add(batch.size());
    } catch (const std::exception &e) {
      S.fail(e.what());
    }
    S.finished(batch.size());
  }
}

// re-plans the block into `store` (plan_jobs) and returns the leaves; with check: the records must equal the dump's own
std::vector<JobId> plan_instance(const qb::Dump &dump, qb::ProofStore &store, bool check) {
  const qb::OpJobIds ops = qb::OpJobIds::dummy_from_config(dump.config);
  const size_t num_input_witnesses = dump.config.add_deposit_count + 1;  // qbench.rs:37
  qb::ProofStore planned;
  const std::vector<JobId> leaves = qb::plan_jobs(planned, ops, num_input_witnesses, dump.config.checkpoint_id);
  if (check) {
    size_t in_dump = 0;
    for (const auto &kv : dump.store.proofs)
      if (kv.first.data_type == qb::Counter) in_dump++;
    for (const auto &kv : planned.proofs) {
      auto it = dump.store.proofs.find(kv.first);
      if (it == dump.store.proofs.end()) die("--check-plan: the dump has no record " + kv.first.hex());
      if (it->second != kv.second) die("--check-plan: record " + kv.first.hex() + " differs from the dump's");
    }
    if (in_dump != planned.proofs.size())
      die("--check-plan: the dump holds " + std::to_string(in_dump) + " counter records, plan_jobs wrote " + std::to_string(planned.proofs.size()));
  }
  for (const auto &kv : planned.proofs) store.proofs[kv.first] = kv.second;
  return leaves;
}

std::string json_escape(const std::string &s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += c; }
    else if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
    else o += c;
  }
  return o;
}

// every proving job of a block, found by walking the planned DAG from its leaves, numbered per circuit type in the order of
// the 24-byte ids: the k-th job of a type proves the k-th witness the pack binds to that type
std::unordered_map<JobId, size_t, qb::JobIdHash> job_ordinals(const qb::ProofStore &store, const std::vector<JobId> &leaves) {
  std::unordered_map<JobId, size_t, qb::JobIdHash> seen;
  std::deque<JobId> q(leaves.begin(), leaves.end());
  std::vector<JobId> proving;
  while (!q.empty()) {
    const JobId j = q.front();
    q.pop_front();
    if (seen.count(j)) continue;
    seen[j] = 0;
    if (j.topic == qb::GenerateStandardProof) proving.push_back(j);
    if (j.topic == qb::NotifyOrchestratorComplete) continue;
    if (store.has(j.next_jobs_id_of_counter()))
      for (const JobId &nj : store.get_next_jobs(j)) q.push_back(nj);
  }
  std::sort(proving.begin(), proving.end(), [](const JobId &a, const JobId &b) {
    return a.circuit_type != b.circuit_type ? a.circuit_type < b.circuit_type : a.hex() < b.hex();
  });
  std::unordered_map<JobId, size_t, qb::JobIdHash> ord;
  size_t k = 0;
  for (size_t i = 0; i < proving.size(); i++) {
    k = (i > 0 && proving[i].circuit_type == proving[i - 1].circuit_type) ? k + 1 : 0;
    ord[proving[i]] = k;
  }
  return ord;
}

int run_qbench(const Options &opt) {
  if (opt.inputs.empty()) die("-i/--input: at least one dump");
  std::vector<qb::Dump> dumps;
  for (const auto &path : opt.inputs) {
    try {
      dumps.push_back(qb::parse_dump(qb::read_file(path)));
    } catch (const std::exception &e) {
      die(path + ": " + e.what());
    }
  }
  // instances in the reference's order: dump by dump, iteration by iteration
  std::vector<std::unique_ptr<Instance>> instances;
  std::vector<std::vector<JobId>> leaves;
  {
    for (size_t d = 0; d < dumps.size(); d++) {
      for (int it = 0; it < opt.iterations; it++) {
        auto inst = std::make_unique<Instance>();
        inst->index = instances.size();
        inst->dump_index = d;
        inst->iteration = it;
        inst->store = dumps[d].store;
        leaves.push_back(plan_instance(dumps[d], inst->store, opt.check_plan));
        instances.push_back(std::move(inst));
      }
    }
  }
  if (opt.ref_counters && opt.blocks_in_flight != 1) die("--ref-counters needs --blocks-in-flight 1 (one store per dump, used iteration after iteration)");

  // circuit pack + workers
  qb::Pack pack;
  Shared shared;
  shared.pack = &pack;
  for (size_t d = 0; d < dumps.size(); d++) {
    size_t first_instance = d * (size_t)opt.iterations;
    shared.ordinals.push_back(job_ordinals(instances[first_instance]->store, leaves[first_instance]));
  }
  size_t oracle_checked = 0, verified = 0;
  std::vector<Worker> workers;
  std::vector<int> devices = opt.devices;
  if (!opt.dry_run) {
    if (opt.pack_dir.empty()) die("--pack DIR is required (circuit files + witnesses; tools/make_circuit_pack.py writes a synthetic one)");
    const int n_dev = cp_device_count();
    if (n_dev <= 0) die("no HIP device visible: this library has no CPU fallback (use --dry-run to exercise the schedule only)");
    if (devices.empty())
      for (int d = 0; d < n_dev; d++) devices.push_back(d);
    for (int d : devices)
      if (d < 0 || d >= n_dev) die("--devices: device " + std::to_string(d) + " is not visible (have " + std::to_string(n_dev) + ")");
    try {
      pack = qb::load_pack(opt.pack_dir);
    } catch (const std::exception &e) {
      die(std::string("circuit pack: ") + e.what());
    }
    const size_t n_general = devices.size() * (size_t)opt.contexts;
    workers.resize(n_general + devices.size() * (size_t)opt.stark_contexts);
    try {
      for (size_t w = 0; w < n_general; w++) {
        workers[w].index = (int)w;
        workers[w].role = opt.stark_contexts > 0 ? 1 : 0;
        workers[w].open(pack, devices[w / (size_t)opt.contexts], opt.lanes);
      }
      for (size_t w = n_general; w < workers.size(); w++) {  // STARK-only contexts: no resident circuits
        workers[w].index = (int)w;
        workers[w].role = 2;
        workers[w].open_bare(devices[(w - n_general) / (size_t)opt.stark_contexts]);
      }
      // warm-up = the gate every distinct proof passes before the clock starts (allocations and the staging ring come with it)
      shared.circuit_class = workers[0].circuit_classes();
      shared.expected.assign(pack.witnesses.size(), {});
      for (auto &w : workers) {
        if (w.role == 2) {
          w.stark.open(w.ctx, opt.stark_log_rows);
          w.has_stark = true;
          w.stark.prove(w.ctx, 0);
          continue;
        }
        w.gate(shared.expected, &oracle_checked, &verified);
        if (opt.groth16_log > 0) {
          w.groth16.open(w.ctx, opt.groth16_log);
          w.has_groth16 = true;
          w.groth16.prove(w.ctx, 0);
        }
        if (opt.stark_log_rows > 0 && opt.stark_contexts == 0) {
          w.stark.open(w.ctx, opt.stark_log_rows);
          w.has_stark = true;
          w.stark.prove(w.ctx, 0);
        }
      }
      for (auto &w : workers) { w.parity_checked = 0; w.proofs = 0; w.launches = 0; }
    } catch (const std::exception &e) {
      die(e.what());
    }
  }
  // --callers T: T threads per context, each the reference's loop (one job per pop, one proof per call), merged by a cp_batcher
  const size_t per_worker = opt.callers > 0 && !opt.dry_run ? (size_t)opt.callers : 1;
  if (opt.callers > 0 && !opt.dry_run) {
    if (opt.groth16_log > 0) die("--callers shares a context between threads: the Groth16 stage (one caller per context) cannot run there");
    if (opt.stark_log_rows > 0) die("--callers shares a context between threads: the STARK stage (one caller per context) cannot run there");
    for (auto &w : workers) {
      w.batcher = cp_batcher_create(w.ctx, (size_t)opt.batch, (unsigned)opt.linger_us);
      if (!w.batcher) die(std::string("cp_batcher_create: ") + cp_last_error(nullptr));
    }
  }
  const size_t n_dev_slots = opt.devices.empty() ? 1 : opt.devices.size();
  const size_t n_stark_only = (opt.dry_run ? n_dev_slots : devices.size()) * (size_t)opt.stark_contexts;  // the LAST workers
  const size_t n_workers = opt.dry_run ? n_dev_slots * (size_t)std::max(1, opt.contexts) + n_stark_only : workers.size() * per_worker;
  auto role_of = [&](size_t w) { return n_stark_only == 0 ? 0 : w >= n_workers - n_stark_only ? 2 : 1; };

  Scheduler S;
  S.stark_stage = opt.stark_log_rows > 0;
  S.roles = S.stark_stage && opt.stark_contexts > 0;
  if (!opt.dry_run)
    for (int t = 0; t < 256; t++) {
      if (!pack.by_type.count(t) && !pack.by_type.count(-1)) continue;
      const auto &st = pack.stages_for((uint8_t)t);
      for (size_t k = 0; k < 8; k++) S.stage_class[t][k] = st.empty() ? 0 : shared.circuit_class[st[k < st.size() ? k : st.size() - 1].circuit];
    }
  // --dry-run --devices a,b,...: the worker -> device assignment of a multi-GPU run without any GPU (one slot per device x context)
  const size_t dry_slots = opt.dry_run && !opt.devices.empty() ? opt.devices.size() * (size_t)opt.contexts : 0;
  std::vector<std::atomic<size_t>> slot_jobs(dry_slots ? dry_slots : 1);
  for (auto &a : slot_jobs) a = 0;
  const double t_begin = now_s();
  size_t next = 0;
  if (opt.sliding) {
    // a WINDOW of --blocks-in-flight blocks: the block that completes starts the next one, so the sequential tail of one block
    // (seven dependent proofs) runs beside the leaves of another — the steady state of a worker that is fed continuously
    if (opt.ref_counters) die("--sliding and --ref-counters do not go together");
    std::atomic<size_t> to_start{0};
    auto start_next = [&] {
      const size_t i = to_start.fetch_add(1);
      if (i >= instances.size()) return;
      Instance &inst = *instances[i];
      inst.t_start = now_s();
      {
        std::lock_guard<std::mutex> l(S.m);
        S.pending_instances++;
      }
      S.enqueue(&inst, leaves[i]);
    };
    S.on_block_complete = start_next;
    S.n_workers = n_workers - n_stark_only;  // the workers that share plonky2 launches
    for (int k = 0; k < opt.blocks_in_flight; k++) start_next();
    std::vector<std::thread> threads;
    for (size_t w = 0; w < n_workers; w++)
      threads.emplace_back([&, w] {
        worker_loop(opt, S, opt.dry_run ? nullptr : &workers[w / per_worker], shared,
                    opt.callers > 0 && !opt.dry_run ? 1 : (size_t)opt.batch, dry_slots && w < dry_slots ? &slot_jobs[w] : nullptr, role_of(w));
      });
    for (auto &t : threads) t.join();
    next = instances.size();
  }
  // (default) instances are started in waves of --blocks-in-flight; the next wave starts when the queue has drained
  while (next < instances.size() && !S.failed) {
    const size_t wave_end = std::min(instances.size(), next + (size_t)opt.blocks_in_flight);
    {
      std::lock_guard<std::mutex> l(S.m);
      S.pending_instances = wave_end - next;
    }
    for (size_t i = next; i < wave_end; i++) {
      Instance &inst = *instances[i];
      if (opt.ref_counters && inst.iteration > 0) {
        // the reference clones the store once per dump and re-plans into it every iteration (qbench.rs:39-50): plan_jobs
        // rewrites the records but `counters` keeps counting
        inst.store = instances[i - 1]->store;
        plan_instance(dumps[inst.dump_index], inst.store, false);
      }
      inst.t_start = now_s();
      S.enqueue(&inst, leaves[i]);
    }
    std::vector<std::thread> threads;
    S.n_workers = n_workers - n_stark_only;  // the workers that share plonky2 launches
    for (size_t w = 0; w < n_workers; w++)
      threads.emplace_back([&, w] {
        worker_loop(opt, S, opt.dry_run ? nullptr : &workers[w / per_worker], shared,
                    opt.callers > 0 && !opt.dry_run ? 1 : (size_t)opt.batch, dry_slots && w < dry_slots ? &slot_jobs[w] : nullptr, role_of(w));
      });
    for (auto &t : threads) t.join();
    if (opt.ref_counters && !S.failed && !instances[next]->complete) {
      // the reference's quirk: later iterations stop after the leaves; the queue simply empties (qbench.rs:52)
      std::lock_guard<std::mutex> l(S.m);
      S.pending_instances = 0;
    }
    next = wave_end;
  }
  const double t_end = now_s();
  if (S.failed && !(opt.ref_counters && S.error.find("ran dry") != std::string::npos)) die(S.error);

  // -o: Vec<QWorkerJobBenchmark> as serde_json::to_vec_pretty writes it (job_id.rs:194-202, qbench.rs:81-82)
  if (!opt.output.empty()) {
    FILE *f = fopen(opt.output.c_str(), "wb");
    if (!f) die("cannot write " + opt.output);
    if (S.benchmarks.empty()) fputs("[]", f);
    else {
      fputs("[\n", f);
      for (size_t i = 0; i < S.benchmarks.size(); i++)
        fprintf(f, "  {\n    \"job_id\": \"%s\",\n    \"duration\": %llu\n  }%s\n", S.benchmarks[i].job.hex().c_str(),
                (unsigned long long)S.benchmarks[i].duration_ms, i + 1 < S.benchmarks.size() ? "," : "");
      fputs("]", f);
    }
    fclose(f);
  }
  if (!opt.trace_path.empty()) {  // one JSON object per line: every job popped, in pop order, then every benchmark with its timing
    FILE *f = fopen(opt.trace_path.c_str(), "wb");
    if (!f) die("cannot write " + opt.trace_path);
    for (const JobId &j : S.processed)
      fprintf(f, "{\"popped\": \"%s\", \"topic\": %u, \"circuit_type\": %u, \"group_id\": %u, \"sub_group_id\": %u, \"task_index\": %u}\n", j.hex().c_str(),
              j.topic, j.circuit_type, j.group_id, j.sub_group_id, j.task_index);
    for (const auto &b : S.benchmarks)
      fprintf(f, "{\"job_id\": \"%s\", \"circuit_type\": %u, \"start_ms\": %.3f, \"end_ms\": %.3f, \"worker\": %d, \"batch\": %d}\n", b.job.hex().c_str(),
              b.job.circuit_type, (b.t0 - t_begin) * 1e3, (b.t1 - t_begin) * 1e3, b.worker, b.batch);
    fclose(f);
  }
  size_t complete = 0, jobs = 0, proofs = 0, parity = 0;
  double latency_sum = 0;
  for (const auto &inst : instances) {
    if (inst->complete) { complete++; latency_sum += inst->t_end - inst->t_start; }
    jobs += inst->jobs_done;
    proofs += inst->proofs_done;
  }
  size_t groth16_proofs = 0, launches = 0, launched = 0, stark_proofs = 0, stark_bytes = 0;
  for (const auto &w : workers) {
    parity += w.parity_checked; groth16_proofs += w.groth16_proofs; launches += w.launches; launched += w.proofs;
    stark_proofs += w.stark_proofs; stark_bytes += w.stark_bytes;
  }
  stark_proofs += S.dry_stark_stages;
  std::string per_device = "null";
  if (dry_slots) {  // jobs taken by the worker slots of each device
    per_device = "{";
    for (size_t d = 0; d < opt.devices.size(); d++) {
      size_t n = 0;
      for (int c = 0; c < opt.contexts; c++) n += slot_jobs[d * (size_t)opt.contexts + (size_t)c].load();
      per_device += (d ? ", \"" : "\"") + std::to_string(opt.devices[d]) + "\": " + std::to_string(n);
    }
    per_device += "}";
  }
  int n_classes = 0;
  for (int c : shared.circuit_class) n_classes = std::max(n_classes, c + 1);
  const double wall = t_end - t_begin;
  std::string devs;
  for (size_t i = 0; i < devices.size(); i++) devs += (i ? "," : "") + std::to_string(devices[i]);
  printf("{\"harness\": \"cityprover-qbench\", \"mode\": \"%s\", \"dumps\": %zu, \"iterations\": %d, \"blocks\": %zu, \"blocks_complete\": %zu, "
         "\"jobs\": %zu, \"proofs\": %zu, \"jobs_per_block\": %.1f, \"proofs_per_block\": %.1f, \"wall_s\": %.6f, \"blocks_per_s\": %.4f, "
         "\"proofs_per_s\": %.2f, \"mean_block_latency_ms\": %.2f, \"devices\": [%s], \"contexts_per_device\": %d, \"workers\": %zu, "
         "\"callers_per_context\": %d, \"lanes_per_context\": %d, \"linger_us\": %d, \"max_batch\": %d, \"blocks_in_flight\": %d, \"window\": \"%s\", \"proofs_byte_checked\": %zu, \"distinct_proofs\": %zu, \"distinct_proofs_equal_to_recorded_bytes\": %zu, \"distinct_proofs_cp_verified\": %zu, "
         "\"circuits\": %zu, \"witnesses\": %zu, \"batch_classes\": %d, \"launches\": %zu, \"mean_batch\": %.2f, \"dry_run_jobs_per_device\": %s, "
         "\"groth16_proofs\": %zu, \"groth16_log_constraints\": %d, \"stark_proofs\": %zu, \"stark_log_rows\": %d, \"stark_proof_bytes_mean\": %.0f, \"pack\": \"%s\", \"timed\": \"from the first enqueue to the "
         "last completion; circuits resident, witnesses page-locked on the host (PCIe-inclusive), witness generation excluded\"}\n",
         opt.dry_run ? "dry-run" : "qbench", dumps.size(), opt.iterations, instances.size(), complete, jobs, proofs,
         instances.empty() ? 0.0 : (double)jobs / instances.size(), instances.empty() ? 0.0 : (double)proofs / instances.size(), wall,
         wall > 0 ? complete / wall : 0.0, wall > 0 ? proofs / wall : 0.0, complete ? latency_sum / complete * 1e3 : 0.0, devs.c_str(), opt.contexts,
         n_workers, opt.dry_run ? 0 : opt.callers, opt.lanes, opt.linger_us, opt.batch, opt.blocks_in_flight, opt.sliding ? "sliding" : "waves", parity, oracle_checked + verified,
         oracle_checked, verified, pack.circuit_files.size(), pack.witnesses.size(), n_classes, launches, launches ? (double)launched / launches : 0.0,
         per_device.c_str(), groth16_proofs, opt.groth16_log, stark_proofs, opt.stark_log_rows, stark_proofs ? (double)stark_bytes / (double)stark_proofs : 0.0,
         json_escape(opt.pack_dir).c_str());
  for (auto &w : workers) w.close();
  return complete == instances.size() || opt.ref_counters ? 0 : 1;
}

// raw throughput: every worker proves --iters batches of --batch proofs, cycling through the pack's bindings
int run_throughput(const Options &opt) {
  if (opt.pack_dir.empty()) die("--pack DIR is required");
  const int n_dev = cp_device_count();
  if (n_dev <= 0) die("no HIP device visible: this library has no CPU fallback");
  std::vector<int> devices = opt.devices;
  if (devices.empty())
    for (int d = 0; d < n_dev; d++) devices.push_back(d);
  qb::Pack pack;
  try {
    pack = qb::load_pack(opt.pack_dir);
  } catch (const std::exception &e) {
    die(std::string("circuit pack: ") + e.what());
  }
  std::vector<Worker> workers(devices.size() * (size_t)opt.contexts);
  std::vector<std::vector<uint8_t>> expected(pack.witnesses.size());
  std::vector<Worker::Item> pairs;  // every distinct (circuit, witness) of the pack's largest batch-compatibility class
  size_t oracle_checked = 0, verified = 0;
  try {
    for (size_t w = 0; w < workers.size(); w++) {
      workers[w].index = (int)w;
      workers[w].open(pack, devices[w / (size_t)opt.contexts], opt.lanes);
      if (!opt.skip_gate) workers[w].gate(expected, &oracle_checked, &verified);
    }
    const std::vector<int> cls = workers[0].circuit_classes();
    std::vector<size_t> per_class;
    for (size_t wi = 0; wi < pack.witnesses.size(); wi++) {
      const int c = cls[pack.witness_circuit[wi]];
      if ((size_t)c >= per_class.size()) per_class.resize(c + 1, 0);
      per_class[c]++;
    }
    const int best = (int)(std::max_element(per_class.begin(), per_class.end()) - per_class.begin());
    for (size_t wi = 0; wi < pack.witnesses.size(); wi++)
      if (cls[pack.witness_circuit[wi]] == best) pairs.push_back({pack.witness_circuit[wi], (int)wi});
    for (auto &w : workers) {
      std::vector<Worker::Item> full;
      for (int j = 0; j < opt.batch; j++) full.push_back(pairs[(size_t)j % pairs.size()]);
      w.prove_items(full, &expected);  // the staging buffers of the full batch size
      w.parity_checked = 0;
    }
  } catch (const std::exception &e) {
    die(e.what());
  }
  std::mutex m;
  std::condition_variable cv;
  size_t waiting = 0;
  bool go = false;
  std::string error;
  std::vector<std::thread> threads;
  for (size_t w = 0; w < workers.size(); w++)
    threads.emplace_back([&, w] {
      {
        std::unique_lock<std::mutex> l(m);
        waiting++;
        cv.notify_all();
        cv.wait(l, [&] { return go; });
      }
      try {
        for (int it = 0; it < opt.iters; it++) {  // a batch = opt.batch consecutive distinct proofs of the pack, a different window every time
          std::vector<Worker::Item> items;
          for (int j = 0; j < opt.batch; j++) items.push_back(pairs[((w * (size_t)opt.iters + (size_t)it) * (size_t)opt.batch + (size_t)j) % pairs.size()]);
          workers[w].prove_items(items, &expected);
        }
      } catch (const std::exception &e) {
        std::lock_guard<std::mutex> l(m);
        error = e.what();
      }
    });
  {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return waiting == workers.size(); });
    go = true;
    cv.notify_all();
  }
  const double t0 = now_s();
  for (auto &t : threads) t.join();
  const double dt = now_s() - t0;
  if (!error.empty()) die(error);
  const size_t proofs = workers.size() * (size_t)opt.iters * (size_t)opt.batch;
  size_t parity = 0;
  for (const auto &w : workers) parity += w.parity_checked;
  printf("{\"harness\": \"cityprover-qbench\", \"mode\": \"throughput\", \"devices\": %zu, \"contexts_per_device\": %d, \"lanes_per_context\": %d, "
         "\"max_batch\": %d, \"proofs\": %zu, \"wall_s\": %.6f, \"proofs_per_s\": %.2f, \"blocks_per_s\": %.3f, \"proofs_byte_checked\": %zu, "
         "\"distinct_proofs\": %zu, \"distinct_proofs_equal_to_recorded_bytes\": %zu, \"distinct_proofs_cp_verified\": %zu, "
         "\"wires\": \"host (page-locked), PCIe-inclusive\"}\n",
         devices.size(), opt.contexts, opt.lanes, opt.batch, proofs, dt, proofs / dt, proofs / dt / 64.0, parity, pairs.size(), oracle_checked, verified);
  for (auto &w : workers) w.close();
  return 0;
}

// one-job-at-a-time callers: --callers threads, each proving --iters single proofs through a cp_batcher (one per context) —
// the reference's worker loops (actors/simple.rs:32-56) as threads of one process sharing a GPU
int run_callers(const Options &opt_in) {
  Options opt = opt_in;
  if (opt.callers < 1) opt.callers = 32;
  if (opt.pack_dir.empty()) die("--pack DIR is required");
  const int n_dev = cp_device_count();
  if (n_dev <= 0) die("no HIP device visible: this library has no CPU fallback");
  std::vector<int> devices = opt.devices;
  if (devices.empty())
    for (int d = 0; d < n_dev; d++) devices.push_back(d);
  qb::Pack pack;
  try {
    pack = qb::load_pack(opt.pack_dir);
  } catch (const std::exception &e) {
    die(std::string("circuit pack: ") + e.what());
  }
  std::vector<Worker::Item> bindings;  // every distinct (circuit, witness) pair of the pack
  for (size_t wi = 0; wi < pack.witnesses.size(); wi++) bindings.push_back({pack.witness_circuit[wi], (int)wi});
  std::vector<Worker> workers(devices.size() * (size_t)opt.contexts);
  std::vector<cp_batcher *> batchers(workers.size(), nullptr);
  std::vector<std::vector<uint8_t>> expected(pack.witnesses.size());
  try {
    for (size_t w = 0; w < workers.size(); w++) {
      workers[w].index = (int)w;
      workers[w].open(pack, devices[w / (size_t)opt.contexts], opt.lanes);
      workers[w].gate(expected, nullptr, nullptr);   // recorded bytes or cp_verify: what the callers' proofs are compared with
      batchers[w] = cp_batcher_create(workers[w].ctx, (size_t)opt.batch, (unsigned)opt.linger_us);
      if (!batchers[w]) throw std::runtime_error(std::string("cp_batcher_create: ") + cp_last_error(nullptr));
    }
  } catch (const std::exception &e) {
    die(e.what());
  }
  std::mutex m;
  std::string error;
  std::atomic<size_t> parity{0};
  auto caller = [&](size_t t, int iters) {
    Worker &wk = workers[t % workers.size()];
    try {
      for (int it = 0; it < iters; it++) {
        const Worker::Item &b = bindings[(t + (size_t)it) % bindings.size()];
        const qb::Witness &wt = *pack.witnesses[b.witness];
        uint8_t *out = nullptr;
        size_t len = 0;
        if (cp_batcher_prove(batchers[t % workers.size()], wk.circuits[b.circuit], wk.wires[b.witness], wt.public_inputs.data(),
                             wt.public_inputs.size(), 0, 0, &out, &len) != CP_OK)
          throw std::runtime_error(std::string("cp_batcher_prove: ") + cp_last_error(nullptr));
        const std::vector<uint8_t> &want = expected[b.witness];
        const bool same = len == want.size() && memcmp(out, want.data(), len) == 0;
        cp_free(out);
        if (!same) throw std::runtime_error("proof bytes differ from the bytes that passed the gate for this witness");
        parity++;
      }
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> l(m);
      error = e.what();
    }
  };
  auto run = [&](int iters) {
    std::vector<std::thread> threads;
    for (size_t t = 0; t < (size_t)opt.callers; t++) threads.emplace_back(caller, t, iters);
    for (auto &t : threads) t.join();
    if (!error.empty()) die(error);
  };
  run(std::min((int)bindings.size(), 8));  // warm-up: staging buffers of the batch sizes met
  std::vector<cp_batcher_stats> before(workers.size());
  for (size_t w = 0; w < workers.size(); w++) cp_batcher_get_stats(batchers[w], &before[w]);
  parity = 0;
  const double t0 = now_s();
  run(opt.iters);
  const double dt = now_s() - t0;
  uint64_t calls = 0, batches = 0, largest = 0, retried = 0;
  for (size_t w = 0; w < workers.size(); w++) {
    cp_batcher_stats st;
    cp_batcher_get_stats(batchers[w], &st);
    calls += st.calls - before[w].calls;
    batches += st.batches - before[w].batches;
    retried += st.retried_singly - before[w].retried_singly;
    if (st.largest_batch > largest) largest = st.largest_batch;
    cp_batcher_destroy(batchers[w]);
  }
  printf("{\"harness\": \"cityprover-qbench\", \"mode\": \"callers\", \"devices\": %zu, \"contexts_per_device\": %d, \"lanes_per_context\": %d, "
         "\"callers\": %d, \"max_batch\": %d, \"linger_us\": %d, \"proofs\": %llu, \"batches\": %llu, \"mean_batch\": %.2f, \"largest_batch\": %llu, "
         "\"retried_singly\": %llu, \"wall_s\": %.6f, \"proofs_per_s\": %.2f, \"blocks_per_s\": %.3f, \"proofs_byte_checked\": %zu, "
         "\"wires\": \"host (page-locked), PCIe-inclusive; every caller proves one job per call\"}\n",
         devices.size(), opt.contexts, opt.lanes, opt.callers, opt.batch, opt.linger_us, (unsigned long long)calls, (unsigned long long)batches,
         batches ? (double)calls / batches : 0.0, (unsigned long long)largest, (unsigned long long)retried, dt, calls / dt, calls / dt / 64.0, parity.load());
  for (auto &w : workers) w.close();
  return 0;
}

// ---- a worker of a LIVE deployment: `city-rollup-cli l2-worker` (city_rollup_core_worker/src/lib.rs:104-146) on the GPU ----------
// The reference's loop, job for job: RSMQ pop_message("JOB") -> serde_json QProvingJobDataID (event_processor.rs:30-40) ->
// process_job (actors/simple.rs:57-115) against the Redis proof store (city_redis_store/src/lib.rs:53-112) -> HINCRBY the
// group counter and, at the goal, send_message the next jobs; NotifyOrchestratorComplete -> a CoreJobCompleted message on
// NOTIFICATIONS. Circuits and witnesses come from the pack, as everywhere in this harness (CircuitData cannot be built, and
// witness generation is outside the build); the job's INPUTS are checked in the store exactly as the dump replay checks them.
// --dry-run proves nothing and needs no GPU (tests/test_qbench_redis.py runs it against an in-process fake server).
int run_redis_worker(const Options &opt) {
  if (opt.redis_uri.empty()) die("--redis HOST:PORT is required");
  qb::RespClient conn;
  conn.connect(opt.redis_uri);
  qb::RedisStore store(conn);
  qb::RsmqQueue queue(conn);
  qb::Pack pack;
  Worker worker;
  std::vector<std::vector<uint8_t>> expected;
  if (!opt.dry_run) {
    if (opt.pack_dir.empty()) die("--pack DIR is required");
    if (cp_device_count() <= 0) die("no HIP device visible: this library has no CPU fallback (use --dry-run to exercise the loop only)");
    pack = qb::load_pack(opt.pack_dir);
    worker.open(pack, opt.devices.empty() ? 0 : opt.devices[0], opt.lanes);
    expected.assign(pack.witnesses.size(), {});
    worker.gate(expected, nullptr, nullptr);
  }
  size_t jobs = 0, proving_jobs = 0, proofs = 0, released = 0, notifications = 0, polls = 0, rounds = 0, launches = 0;
  // --redis-batch N (default 1: the reference's loop, one job per pop): the worker keeps up to N jobs open. A round (1) takes
  // messages that are in the queue NOW until N jobs are open, (2) proves ONE stage of every open job — one launch per
  // batch-compatibility class, (3) does the bookkeeping of the jobs whose last stage that was (output, counter, next jobs: they
  // can be in the very next round's launch) and keeps the others for their next stage. A job is only ever in the queue after
  // everything it depends on has been stored, so open jobs are independent of each other.
  const size_t take = opt.redis_batch > 1 ? (size_t)opt.redis_batch : 1;
  std::vector<int> cls;
  if (!opt.dry_run) cls = worker.circuit_classes();
  struct Open { JobId job; int stage; };
  std::vector<Open> open;
  // the tail of process_job (actors/simple.rs:89-115) for one job
  auto complete = [&](const JobId &job, std::vector<uint8_t> output) {
    jobs++;
    if (job.topic == qb::GenerateStandardProof) {
      if (job.circuit_type == qb::WrapFinalSigHashProofBLS12381) output = zero_groth16_bincode();  // GROTH16_DISABLED_DEV_MODE
      store.set_bytes(job.output_id(), output);
      proving_jobs++;
      proofs += (size_t)qb::proofs_per_job(job.circuit_type);
    }
    if (job.topic == qb::NotifyOrchestratorComplete) {
      queue.send("NOTIFICATIONS", "0");  // serde_json of QueueNotification::CoreJobCompleted (serde_repr u8)
      notifications++;
      return;
    }
    const uint32_t goal = store.get_goal(job);
    if (goal != 0 && store.inc_counter(job.counter_id()) == goal)
      for (const JobId &nj : store.get_next_jobs(job)) {
        queue.send("JOB", qb::job_to_json(nj));
        released++;
      }
  };
  size_t taken = 0;
  const double t0 = now_s();
  try {
  for (;;) {
    while (open.size() < take && !(opt.max_jobs > 0 && taken >= (size_t)opt.max_jobs)) {
      std::string body;
      if (!queue.pop("JOB", body)) break;
      const JobId job = qb::job_from_json(body);
      taken++;
      if (job.topic != qb::GenerateStandardProof) {  // barrier and notify jobs: nothing to prove
        complete(job, {});
        continue;
      }
      const std::vector<uint8_t> w = store.get_bytes(job);  // the witness must be there, and every proof it names
      for (const JobId &dep : qb::proof_dependencies(job, w))
        if (store.get_bytes(dep).empty()) throw qb::StoreError("Proof " + dep.hex() + " needed by " + job.hex() + " is empty");
      if (!opt.dry_run && (int)pack.stages_for(job.circuit_type).size() != qb::proofs_per_job(job.circuit_type))
        throw std::runtime_error("the pack binds the wrong number of stages to circuit type " + std::to_string(job.circuit_type));
      open.push_back({job, 0});
    }
    if (open.empty()) {
      if (opt.drain || (opt.max_jobs > 0 && taken >= (size_t)opt.max_jobs)) {
        if (queue.size("JOB") == 0 || (opt.max_jobs > 0 && taken >= (size_t)opt.max_jobs)) break;
        continue;  // barrier jobs released more work while this round was taking messages
      }
      polls++;
      std::this_thread::sleep_for(std::chrono::milliseconds(250));  // event_processor.rs:36
      continue;
    }
    rounds++;
    std::vector<std::vector<uint8_t>> stage_out(open.size(), std::vector<uint8_t>(1, 0));
    if (!opt.dry_run) {
      std::vector<std::pair<int, size_t>> order;  // (class of the stage's circuit, open job)
      for (size_t i = 0; i < open.size(); i++) order.push_back({cls[pack.stages_for(open[i].job.circuit_type)[(size_t)open[i].stage].circuit], i});
      std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
      for (size_t lo = 0; lo < order.size();) {
        size_t hi = lo;
        std::vector<Worker::Item> items;
        while (hi < order.size() && order[hi].first == order[lo].first) {
          const Open &o = open[order[hi].second];
          const qb::Binding &b = pack.stages_for(o.job.circuit_type)[(size_t)o.stage];
          items.push_back({b.circuit, b.witness_for(o.job.task_index)});
          hi++;
        }
        auto pr = worker.prove_items(items, &expected);
        launches++;
        for (size_t k = lo; k < hi; k++) stage_out[order[k].second] = std::move(pr[k - lo]);
        lo = hi;
      }
    }
    std::vector<Open> still;
    for (size_t i = 0; i < open.size(); i++) {
      if (open[i].stage + 1 == qb::proofs_per_job(open[i].job.circuit_type)) {
        const JobId job = open[i].job;
        open[i].stage = -1;  // from here on this job is never re-queued: its bookkeeping runs once, whatever happens in it
        complete(job, std::move(stage_out[i]));
      } else still.push_back({open[i].job, open[i].stage + 1});
    }
    open.swap(still);
  }
  } catch (...) {
    // With --redis-batch N up to N jobs have been popped (destructively) when something fails. The reference's loop loses the ONE
    // job it was working on (actors/simple.rs:32-56: pop, process, error out); so does this one: the jobs that are merely open go
    // back to the queue before the worker exits, best effort (the failure may be the connection itself).
    for (const Open &o : open) {
      if (o.stage < 0) continue;  // completed (or failed while completing) in this round
      try { queue.send("JOB", qb::job_to_json(o.job)); } catch (...) { break; }
    }
    throw;
  }
  const double wall = now_s() - t0;
  printf("{\"harness\": \"cityprover-qbench\", \"mode\": \"redis-worker\", \"dry_run\": %s, \"redis\": \"%s\", \"jobs\": %zu, \"proving_jobs\": %zu, "
         "\"proofs\": %zu, \"jobs_released\": %zu, \"notifications\": %zu, \"idle_polls\": %zu, \"queue_left\": %lld, \"redis_batch\": %zu, \"rounds\": %zu, \"launches\": %zu, \"wall_s\": %.6f, \"proofs_per_s\": %.2f}\n",
         opt.dry_run ? "true" : "false", json_escape(opt.redis_uri).c_str(), jobs, proving_jobs, proofs, released, notifications, polls, queue.size("JOB"), take,
         rounds, launches, wall, wall > 0 ? proofs / wall : 0.0);
  if (!opt.dry_run) worker.close();
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  Options opt;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
    if (a == "-i" || a == "--input") {  // num_args = 1.. (args.rs:106): every following non-option is an input
      opt.inputs.push_back(val());
      while (i + 1 < argc && argv[i + 1][0] != '-') opt.inputs.push_back(argv[++i]);
    } else if (a == "-o" || a == "--output") opt.output = val();
    else if (a == "-n" || a == "--num-iterations") opt.iterations = atoi(val().c_str());
    else if (a == "--network") opt.network = val();  // selects the network magic baked into the circuits: the pack's business here
    else if (a == "--pack") opt.pack_dir = val();
    else if (a == "--mode") opt.mode = val();
    else if (a == "--contexts") opt.contexts = atoi(val().c_str());
    else if (a == "--batch") opt.batch = atoi(val().c_str());
    else if (a == "--iters") opt.iters = atoi(val().c_str());
    else if (a == "--blocks-in-flight") opt.blocks_in_flight = atoi(val().c_str());
    else if (a == "--lanes") opt.lanes = atoi(val().c_str());
    else if (a == "--callers") opt.callers = atoi(val().c_str());
    else if (a == "--linger-us") opt.linger_us = atoi(val().c_str());
    else if (a == "--trace") opt.trace_path = val();
    else if (a == "--groth16-log-size") opt.groth16_log = atoi(val().c_str());
    else if (a == "--stark-log-rows") opt.stark_log_rows = atoi(val().c_str());
    else if (a == "--stark-contexts") opt.stark_contexts = atoi(val().c_str());
    else if (a == "--dry-run") opt.dry_run = true;
    else if (a == "--skip-gate") opt.skip_gate = true;
    else if (a == "--sliding") opt.sliding = true;
    else if (a == "--redis") opt.redis_uri = val();
    else if (a == "--drain") opt.drain = true;
    else if (a == "--max-jobs") opt.max_jobs = atoi(val().c_str());
    else if (a == "--redis-batch") opt.redis_batch = atoi(val().c_str());
    else if (a == "--dry-run-job-us") opt.dry_job_us = atoi(val().c_str());
    else if (a == "--dry-run-stages") opt.dry_stages = true;
    else if (a == "--ref-counters") opt.ref_counters = true;
    else if (a == "--check-plan") opt.check_plan = true;
    else if (a == "--devices") {
      const std::string v = val();
      if (v != "all") {
        size_t p = 0;
        while (p < v.size()) {
          size_t q = v.find(',', p);
          if (q == std::string::npos) q = v.size();
          opt.devices.push_back(atoi(v.substr(p, q - p).c_str()));
          p = q + 1;
        }
      }
    } else die("unknown argument " + a);
  }
  if (opt.groth16_log != 0 && (opt.groth16_log < 4 || opt.groth16_log > 26)) die("--groth16-log-size must be 0 (off) or 4..26");
  if (opt.stark_log_rows != 0 && (opt.stark_log_rows < 6 || opt.stark_log_rows > 20)) die("--stark-log-rows must be 0 (off) or 6..20");
  if (opt.stark_contexts < 0 || opt.stark_contexts > 8 || (opt.stark_contexts > 0 && opt.stark_log_rows == 0)) die("--stark-contexts must be 0..8 and needs --stark-log-rows");
  if (opt.iterations < 1 || opt.contexts < 1 || opt.batch < 1 || opt.blocks_in_flight < 1 || opt.iters < 1 || opt.lanes < 1 || opt.callers < 0 || opt.linger_us < 0) die("bad argument value");
  // Every context owns a HIP stream, and the runtime multiplexes the streams of ONE process onto GPU_MAX_HW_QUEUES hardware
  // queues (default 4): with more contexts than that, kernels of different contexts queue behind each other instead of
  // overlapping (measured: 8 contexts x batch 1 = 656 proofs/s on 4 queues, 930 on 8, 1 020 with 12 x 12). The reference's
  // deployment — one worker PROCESS per job stream — has a runtime and queues per process and does not meet this limit;
  // a pool of threads in one process does. Ask for as many queues as contexts (up to 12) unless the caller set the variable;
  // it is read when the HIP runtime initialises, i.e. at the first cp_* call below.
  const int streams = opt.contexts * (opt.lanes > 1 ? opt.lanes + 1 : 1);  // a lane is a context of its own, beside its parent
  if (streams > 4 && !opt.dry_run) setenv("GPU_MAX_HW_QUEUES", std::to_string(streams < 12 ? streams : 12).c_str(), 0);
  try {
    if (opt.mode == "qbench") return run_qbench(opt);
    if (opt.mode == "throughput") return run_throughput(opt);
    if (opt.mode == "callers") return run_callers(opt);
    if (opt.mode == "redis-worker") return run_redis_worker(opt);
  } catch (const std::exception &e) {
    die(e.what());
  }
  die("--mode must be qbench, throughput, callers or redis-worker");
}
