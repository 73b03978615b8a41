// Job ids, the in-memory proof store, the dump reader and the job planner of the q-bench harness — the data model the
// reference's worker loop runs on, restated for the native harness (tools/cityprover_qbench.cpp). Plain C++17, no GPU.
//   QProvingJobDataID (24 bytes)          city_rollup_common/src/qworker/job_id.rs:204-275
//   SimpleProofStoreMemory                city_rollup_common/src/qworker/memory_proof_store/mod.rs:10-102
//   counter / goal / next-jobs triplets   city_rollup_common/src/qworker/proof_store.rs:12-87
//   BlockProofStoreDump (bincode 1.3)     city_rollup_core_worker_qbench/src/dump.rs:15-26
//   plan_jobs                             city_rollup_core_orchestrator/src/debug/scenario/actors/job_planner.rs:5-154
//   dummy tree-prover ids                 .../block_planner/tree_helper.rs:22-70, transition.rs:110-157
//   BinaryTreePlanner                     city_common/src/tree_planner.rs:61-84
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace qb {

// job_id.rs:10-17, 45-51, 87-125
enum Topic : uint8_t { GenerateStandardProof = 0, GenerateGroth16Proof = 1, BlockUserSignatureProof = 2, NotifyOrchestratorComplete = 3, AggregateJobs = 4 };
enum DataType : uint8_t { InputWitness = 0, BaseInputProof = 1, OutputProof = 8, Counter = 16 };
enum CircuitType : uint8_t {
  RegisterUser = 0, RegisterUserAggregate = 1, AddL1Deposit = 2, AddL1DepositAggregate = 3, ClaimL1Deposit = 4, ClaimL1DepositAggregate = 5,
  TransferTokensL2 = 6, TransferTokensL2Aggregate = 7, AddL1Withdrawal = 8, AddL1WithdrawalAggregate = 9, ProcessL1Withdrawal = 10,
  ProcessL1WithdrawalAggregate = 11, GenerateRollupStateTransitionProof = 32, GenerateSigHashIntrospectionProof = 33,
  GenerateFinalSigHashProof = 34, GenerateFinalSigHashProofGroth16 = 35, WrapFinalSigHashProofBLS12381 = 36,
  AggUserRegisterClaimDepositL2Transfer = 40, AggAddProcessL1WithdrawalAddL1Deposit = 41, DummyRegisterUserAggregate = 48,
  DummyAddL1DepositAggregate = 49, DummyClaimL1DepositAggregate = 50, DummyTransferTokensL2Aggregate = 51,
  DummyAddL1WithdrawalAggregate = 52, DummyProcessL1WithdrawalAggregate = 53, WrappedSignatureProof = 64, Secp256K1SignatureProof = 65,
  UnknownCircuit = 255
};

struct JobId {  // job_id.rs:204-214; the wire form is exactly these 24 bytes (test at job_id.rs:599-615)
  uint8_t topic = 0;
  uint64_t goal_id = 0;
  uint8_t circuit_type = 0;
  uint32_t group_id = 0, sub_group_id = 0, task_index = 0;
  uint8_t data_type = 0, data_index = 0;

  std::array<uint8_t, 24> bytes() const {  // job_id.rs:215-229
    std::array<uint8_t, 24> b{};
    b[0] = topic;
    memcpy(&b[1], &goal_id, 8);
    b[9] = circuit_type;
    memcpy(&b[10], &group_id, 4);
    memcpy(&b[14], &sub_group_id, 4);
    memcpy(&b[18], &task_index, 4);
    b[22] = data_type;
    b[23] = data_index;
    return b;
  }
  static JobId from_bytes(const uint8_t *b) {  // job_id.rs:230-255 (the enum range checks are the caller's: parse())
    JobId j;
    j.topic = b[0];
    memcpy(&j.goal_id, b + 1, 8);
    j.circuit_type = b[9];
    memcpy(&j.group_id, b + 10, 4);
    memcpy(&j.sub_group_id, b + 14, 4);
    memcpy(&j.task_index, b + 18, 4);
    j.data_type = b[22];
    j.data_index = b[23];
    return j;
  }
  bool operator==(const JobId &o) const { return bytes() == o.bytes(); }
  std::string hex() const {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (uint8_t c : bytes()) { s += d[c >> 4]; s += d[c & 15]; }
    return s;
  }
  // job_id.rs:546-577
  JobId output_id() const { JobId j = *this; j.data_type = OutputProof; j.data_index = 0; return j; }
  JobId counter_id() const { JobId j = *this; j.data_type = Counter; j.task_index = 0; j.data_index = 0; return j; }
  JobId goal_id_of_counter() const { JobId j = counter_id(); j.data_index = 1; return j; }
  JobId next_jobs_id_of_counter() const { JobId j = counter_id(); j.data_index = 2; return j; }
  JobId with_task_index(uint32_t t) const { JobId j = *this; j.task_index = t; return j; }
  // job_id.rs:480-545: the aggregation job one level up that consumes this proof
  JobId tree_parent_proof_input_id() const {
    JobId j = *this;
    uint8_t c = circuit_type;
    if (c <= ProcessL1WithdrawalAggregate) c |= 1;  // X -> XAggregate, XAggregate -> XAggregate (types 0..11 come in pairs)
    else if (c >= DummyRegisterUserAggregate && c <= DummyProcessL1WithdrawalAggregate) {
      static const uint8_t parent[6] = {RegisterUserAggregate, AddL1DepositAggregate, ClaimL1DepositAggregate, TransferTokensL2Aggregate,
                                        AddL1WithdrawalAggregate, ProcessL1WithdrawalAggregate};
      c = parent[c - DummyRegisterUserAggregate];
    }
    j.circuit_type = c;
    j.data_type = InputWitness;
    j.data_index = 0;
    j.sub_group_id = sub_group_id + 1;
    j.task_index = task_index >> 1;
    return j;
  }
  static uint32_t circuit_group_id(uint8_t circuit_type) { return (uint32_t)circuit_type + 0xCF00u; }  // job_id.rs:131-133
  static JobId proof_job(uint64_t goal, uint8_t ct, uint32_t group, uint32_t sub, uint32_t task) {  // new_proof_job_id, job_id.rs:340-357
    JobId j;
    j.topic = GenerateStandardProof; j.goal_id = goal; j.circuit_type = ct; j.group_id = group; j.sub_group_id = sub; j.task_index = task;
    return j;
  }
  static JobId core_op_witness(uint8_t ct, uint64_t checkpoint, uint32_t task) { return proof_job(checkpoint, ct, circuit_group_id(ct), 0, task); }
  static JobId aggregate_jobs_group(uint64_t block, uint32_t group, uint32_t task) {  // job_id.rs:376-388
    JobId j;
    j.topic = AggregateJobs; j.goal_id = block; j.circuit_type = UnknownCircuit; j.group_id = group; j.task_index = task;
    return j;
  }
  static JobId notify_block_complete(uint64_t block) {  // job_id.rs:389-400
    JobId j;
    j.topic = NotifyOrchestratorComplete; j.goal_id = block; j.circuit_type = UnknownCircuit;
    return j;
  }
  // job_id.rs:401-479
  static JobId root_job(uint64_t block, uint8_t ct, uint32_t sub, uint32_t task) { return proof_job(block, ct, circuit_group_id(ct), sub, task); }
};

struct JobIdHash {
  size_t operator()(const JobId &j) const {
    const auto b = j.bytes();
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint8_t c : b) { h ^= c; h *= 0x100000001b3ull; }
    return (size_t)h;
  }
};

inline bool valid_enums(const JobId &j) {  // the TryFrom<u8> impls of job_id.rs
  const uint8_t c = j.circuit_type;
  const bool ct = c <= 11 || (c >= 32 && c <= 36) || c == 40 || c == 41 || (c >= 48 && c <= 53) || c == 64 || c == 65 || c == 255;
  const bool dt = j.data_type == 0 || j.data_type == 1 || j.data_type == 8 || j.data_type == 16;
  return j.topic <= 4 && ct && dt;
}

struct StoreError : std::runtime_error { using std::runtime_error::runtime_error; };

// SimpleProofStoreMemory: `proofs` holds witnesses, proofs AND the goal / next-jobs records; `counters` only the running counts
struct ProofStore {
  std::unordered_map<JobId, std::vector<uint8_t>, JobIdHash> proofs;
  std::unordered_map<JobId, uint32_t, JobIdHash> counters;

  const std::vector<uint8_t> &get_bytes(const JobId &id) const {  // memory_proof_store/mod.rs:50-63
    auto it = proofs.find(id);
    if (it == proofs.end()) throw StoreError("Data not found. Wanted " + id.hex());
    return it->second;
  }
  bool has(const JobId &id) const { return proofs.count(id) != 0; }
  void set_bytes(const JobId &id, const uint8_t *p, size_t n) { proofs[id].assign(p, p + n); }
  void set_bytes(const JobId &id, const std::vector<uint8_t> &v) { proofs[id] = v; }
  uint32_t inc_counter(const JobId &id) { return ++counters[id]; }  // mod.rs:77-83
  uint32_t get_goal(const JobId &job) const {  // proof_store.rs:15-21
    const auto &g = get_bytes(job.goal_id_of_counter());
    if (g.size() != 4) throw StoreError("goal record of " + job.hex() + " is not 4 bytes");
    uint32_t v;
    memcpy(&v, g.data(), 4);
    return v;
  }
  std::vector<JobId> get_next_jobs(const JobId &job) const {  // proof_store.rs:22-31: bincode Vec<QProvingJobDataID>
    const auto &b = get_bytes(job.next_jobs_id_of_counter());
    uint64_t n = 0;
    if (b.size() < 8) throw StoreError("next-jobs record of " + job.hex() + " is truncated");
    memcpy(&n, b.data(), 8);
    if (b.size() != 8 + 24 * n) throw StoreError("next-jobs record of " + job.hex() + " has the wrong length");
    std::vector<JobId> out(n);
    for (uint64_t i = 0; i < n; i++) out[i] = JobId::from_bytes(b.data() + 8 + 24 * i);
    return out;
  }
  // proof_store.rs:46-58
  void write_next_jobs(const std::vector<JobId> &jobs, const std::vector<JobId> &next) {
    const JobId counter = jobs.at(0).counter_id();
    const uint32_t zero = 0, goal = (uint32_t)jobs.size();
    set_bytes(counter, (const uint8_t *)&zero, 4);
    set_bytes(jobs[0].goal_id_of_counter(), (const uint8_t *)&goal, 4);
    std::vector<uint8_t> v(8 + 24 * next.size());
    const uint64_t n = next.size();
    memcpy(v.data(), &n, 8);
    for (size_t i = 0; i < next.size(); i++) { const auto b = next[i].bytes(); memcpy(v.data() + 8 + 24 * i, b.data(), 24); }
    set_bytes(jobs[0].next_jobs_id_of_counter(), v);
  }
  // proof_store.rs:65-87: level i releases level i + 1, the last level releases `next`
  void write_multidimensional_jobs(const std::vector<std::vector<JobId>> &levels, const std::vector<JobId> &next) {
    for (size_t i = 0; i < levels.size(); i++) write_next_jobs(levels[i], i + 1 == levels.size() ? next : levels[i + 1]);
  }
};

struct DumpConfig {  // dump.rs:15-20 + transition.rs CityOpJobConfig (field order of the bincode)
  uint64_t checkpoint_id = 0;
  uint32_t rpc_node_id = 0;
  uint64_t register_user_count = 0, claim_deposit_count = 0, token_transfer_count = 0, add_withdrawal_count = 0,
           process_withdrawal_count = 0, add_deposit_count = 0;
};
struct Dump {
  DumpConfig config;
  ProofStore store;
};

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

// bincode 1.3 default options: little-endian fixed-width integers, u64 length prefixes (Cargo.toml pins bincode = "=1.3.3")
inline Dump parse_dump(const std::vector<uint8_t> &b) {
  size_t o = 0;
  auto need = [&](size_t n) { if (n > b.size() - o) throw ParseError("dump truncated at byte " + std::to_string(o)); };
  auto u64 = [&]() { need(8); uint64_t v; memcpy(&v, &b[o], 8); o += 8; return v; };
  auto u32 = [&]() { need(4); uint32_t v; memcpy(&v, &b[o], 4); o += 4; return v; };
  Dump d;
  d.config.checkpoint_id = u64();
  d.config.rpc_node_id = u32();
  d.config.register_user_count = u64();
  d.config.claim_deposit_count = u64();
  d.config.token_transfer_count = u64();
  d.config.add_withdrawal_count = u64();
  d.config.process_withdrawal_count = u64();
  d.config.add_deposit_count = u64();
  const uint64_t n = u64();
  if (n > b.size() / 32) throw ParseError("implausible number of store entries");
  for (uint64_t i = 0; i < n; i++) {
    need(24);
    const JobId k = JobId::from_bytes(&b[o]);
    o += 24;
    if (!valid_enums(k)) throw ParseError("store entry " + std::to_string(i) + ": invalid job id " + k.hex());
    const uint64_t len = u64();
    need(len);
    d.store.proofs[k].assign(b.begin() + o, b.begin() + o + len);
    o += len;
  }
  const uint64_t nc = u64();
  if (nc > b.size() / 28) throw ParseError("implausible number of counters");
  for (uint64_t i = 0; i < nc; i++) {
    need(24);
    const JobId k = JobId::from_bytes(&b[o]);
    o += 24;
    d.store.counters[k] = u32();
  }
  if (o != b.size()) throw ParseError("trailing bytes after the dump");
  return d;
}

inline std::vector<uint8_t> read_file(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw ParseError("cannot open " + path);
  std::vector<uint8_t> v;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

// ---- planner ------------------------------------------------------------------------------------------------------

struct TreePos { uint64_t level, index; };
// BinaryTreePlanner::new(n).levels as (left child position) per node — only the left child names the parent's id
inline std::vector<std::vector<TreePos>> binary_tree_left_children(size_t num_leaves) {  // tree_planner.rs:61-84
  struct Node { TreePos pos; };
  std::vector<TreePos> current(num_leaves);
  for (size_t i = 0; i < num_leaves; i++) current[i] = {0, i};
  std::vector<std::vector<TreePos>> levels;
  uint64_t level_index = 1;
  while (current.size() > 1) {
    std::vector<TreePos> next_pos, lefts;
    for (size_t i = 0; i < current.size() / 2; i++) {
      next_pos.push_back({level_index, i});
      lefts.push_back(current[2 * i]);
    }
    levels.push_back(lefts);
    if (current.size() % 2 == 1) next_pos.push_back(current.back());
    current = next_pos;
    level_index++;
  }
  return levels;
}

// tree_helper.rs:22-70
inline std::vector<std::vector<JobId>> dummy_tree_prover_ids_op_circuit(uint8_t circuit_type, uint8_t dummy_type, uint64_t checkpoint, size_t leaf_count) {
  if (leaf_count == 0) return {{JobId::proof_job(checkpoint, dummy_type, 0xDD, 0, 0)}};
  std::vector<std::vector<JobId>> job_ids(1);
  for (size_t i = 0; i < leaf_count; i++) job_ids[0].push_back(JobId::core_op_witness(circuit_type, checkpoint, (uint32_t)i));
  for (const auto &level : binary_tree_left_children(leaf_count)) {
    std::vector<JobId> ids;
    for (const TreePos &left : level) ids.push_back(job_ids.at(left.level).at(left.index).output_id().tree_parent_proof_input_id());
    job_ids.push_back(ids);
  }
  return job_ids;
}

struct OpJobIds {  // CityOpJobIds, transition.rs:110-157
  std::vector<std::vector<JobId>> register_user, claim_deposit, token_transfer, add_withdrawal, process_withdrawal, add_deposit;
  static OpJobIds dummy_from_config(const DumpConfig &c) {
    OpJobIds o;
    o.register_user = dummy_tree_prover_ids_op_circuit(RegisterUser, DummyRegisterUserAggregate, c.checkpoint_id, c.register_user_count);
    o.claim_deposit = dummy_tree_prover_ids_op_circuit(ClaimL1Deposit, DummyClaimL1DepositAggregate, c.checkpoint_id, c.claim_deposit_count);
    o.token_transfer = dummy_tree_prover_ids_op_circuit(TransferTokensL2, DummyTransferTokensL2Aggregate, c.checkpoint_id, c.token_transfer_count);
    o.add_withdrawal = dummy_tree_prover_ids_op_circuit(AddL1Withdrawal, DummyAddL1WithdrawalAggregate, c.checkpoint_id, c.add_withdrawal_count);
    o.process_withdrawal = dummy_tree_prover_ids_op_circuit(ProcessL1Withdrawal, DummyProcessL1WithdrawalAggregate, c.checkpoint_id, c.process_withdrawal_count);
    o.add_deposit = dummy_tree_prover_ids_op_circuit(AddL1Deposit, DummyAddL1DepositAggregate, c.checkpoint_id, c.add_deposit_count);
    return o;
  }
};

// job_planner.rs:5-154 — writes the counter / goal / next-jobs records of one block and returns the leaf jobs in queue order
inline std::vector<JobId> plan_jobs(ProofStore &store, const OpJobIds &ops, size_t num_input_witnesses, uint64_t checkpoint) {
  const JobId root_state_transition = JobId::root_job(checkpoint, GenerateRollupStateTransitionProof, 0, 0);
  std::vector<JobId> agg_jobs_for_inputs;
  for (size_t i = 0; i < num_input_witnesses; i++) agg_jobs_for_inputs.push_back(JobId::aggregate_jobs_group(checkpoint, 1, (uint32_t)i));
  store.write_next_jobs(agg_jobs_for_inputs, {JobId::notify_block_complete(checkpoint)});
  struct PerInput { JobId wrap, final_, introspection; };
  std::vector<PerInput> per_input;
  for (size_t i = 0; i < num_input_witnesses; i++)
    per_input.push_back({JobId::root_job(checkpoint, WrapFinalSigHashProofBLS12381, (uint32_t)i, (uint32_t)i),
                         JobId::root_job(checkpoint, GenerateFinalSigHashProof, (uint32_t)i, (uint32_t)i),
                         JobId::root_job(checkpoint, GenerateSigHashIntrospectionProof, 0, (uint32_t)i)});
  for (size_t i = 0; i < per_input.size(); i++) {
    store.write_next_jobs({per_input[i].wrap}, {agg_jobs_for_inputs[i]});
    store.write_next_jobs({per_input[i].final_}, {per_input[i].wrap});
  }
  const JobId agg_state_root = JobId::aggregate_jobs_group(checkpoint, 5, 0), agg_all_introspections = JobId::aggregate_jobs_group(checkpoint, 5, 1);
  std::vector<JobId> introspection_jobs, final_jobs;
  for (const auto &p : per_input) { introspection_jobs.push_back(p.introspection); final_jobs.push_back(p.final_); }
  store.write_next_jobs(introspection_jobs, {agg_all_introspections});
  store.write_next_jobs({agg_state_root, agg_all_introspections}, final_jobs);
  store.write_next_jobs({root_state_transition}, {agg_state_root});
  const JobId part1_common = JobId::aggregate_jobs_group(checkpoint, 6, 0), part2_common = JobId::aggregate_jobs_group(checkpoint, 6, 1);
  const JobId part1 = JobId::root_job(checkpoint, AggUserRegisterClaimDepositL2Transfer, 0, 0);
  const JobId part2 = JobId::root_job(checkpoint, AggAddProcessL1WithdrawalAddL1Deposit, 0, 0);
  store.write_next_jobs({part1_common, part2_common}, {root_state_transition});
  store.write_next_jobs({part1}, {part1_common});
  store.write_next_jobs({part2}, {part2_common});
  const JobId reg_agg = JobId::aggregate_jobs_group(checkpoint, 11, 0), claim_agg = JobId::aggregate_jobs_group(checkpoint, 11, 1),
              transfer_agg = JobId::aggregate_jobs_group(checkpoint, 11, 2);
  store.write_next_jobs({reg_agg, claim_agg, transfer_agg}, {part1});
  const JobId addw_agg = JobId::aggregate_jobs_group(checkpoint, 12, 0), procw_agg = JobId::aggregate_jobs_group(checkpoint, 12, 1),
              addd_agg = JobId::aggregate_jobs_group(checkpoint, 12, 2);
  store.write_next_jobs({addw_agg, procw_agg, addd_agg}, {part2});
  store.write_multidimensional_jobs(ops.register_user, {reg_agg});
  store.write_multidimensional_jobs(ops.claim_deposit, {claim_agg});
  store.write_multidimensional_jobs(ops.token_transfer, {transfer_agg});
  store.write_multidimensional_jobs(ops.add_withdrawal, {addw_agg});
  store.write_multidimensional_jobs(ops.process_withdrawal, {procw_agg});
  store.write_multidimensional_jobs(ops.add_deposit, {addd_agg});
  std::vector<JobId> leaves = introspection_jobs;
  for (const auto *lv : {&ops.register_user, &ops.claim_deposit, &ops.token_transfer, &ops.add_withdrawal, &ops.process_withdrawal, &ops.add_deposit})
    leaves.insert(leaves.end(), (*lv)[0].begin(), (*lv)[0].end());
  return leaves;
}

// ---- what a job reads besides its own witness: the proofs of other jobs ----------------------------------------------
// Offsets of the QProvingJobDataID fields inside the bincode of the job witnesses (city_rollup_common/src/qworker/
// job_witnesses/{op,agg,sighash}.rs; QHashOut = u64 length 64 + 64 hex characters = 72 bytes, AggStateTransition = 2 of them).
// The worker fetches these proofs from the store before it can fill the witness (e.g. worker/traits.rs:164-202,
// ops/l2_transfer/circuit.rs:265-281, toolbox/root.rs:259-271): a job whose inputs are missing fails, as in the reference.
inline std::vector<JobId> proof_dependencies(const JobId &job, const std::vector<uint8_t> &w) {
  std::vector<JobId> deps;
  auto id_at = [&](size_t off) {
    if (off + 24 > w.size()) throw StoreError("witness of " + job.hex() + " is too short for a proof id at " + std::to_string(off));
    JobId d = JobId::from_bytes(w.data() + off);
    if (!valid_enums(d)) throw StoreError("witness of " + job.hex() + ": invalid proof id at " + std::to_string(off));
    deps.push_back(d);
  };
  const size_t AST = 144;
  switch (job.circuit_type) {
    case RegisterUserAggregate: case AddL1DepositAggregate: case ClaimL1DepositAggregate: case TransferTokensL2Aggregate:
    case AddL1WithdrawalAggregate: case ProcessL1WithdrawalAggregate: {  // CircuitInputWithDependencies: ..., Vec<id> (u64 count first)
      if (w.size() < 8 + 48) throw StoreError("aggregation witness of " + job.hex() + " is too short");
      uint64_t n;
      memcpy(&n, w.data() + w.size() - 56, 8);
      if (n != 2) throw StoreError("aggregation witness of " + job.hex() + " does not end with two dependencies");
      id_at(w.size() - 48);
      id_at(w.size() - 24);
      break;
    }
    case ClaimL1Deposit: case TransferTokensL2: case AddL1Withdrawal: id_at(w.size() - 24); break;  // signature_proof_id
    case AggUserRegisterClaimDepositL2Transfer: id_at(AST); id_at(AST + 24 + 2 * AST); id_at(AST + 24 + 2 * AST + 24 + AST); break;
    case AggAddProcessL1WithdrawalAddL1Deposit: id_at(2 * AST); id_at(2 * AST + 24 + AST); id_at(2 * AST + 24 + AST + 24 + AST); break;
    case GenerateRollupStateTransitionProof: id_at(2 * AST); id_at(2 * AST + 24 + 3 * AST); break;
    case GenerateFinalSigHashProof: id_at(w.size() - 48); id_at(w.size() - 24); break;
    case WrapFinalSigHashProofBLS12381: id_at(0); break;
    default: break;
  }
  for (JobId &d : deps)
    if (d.data_type == InputWitness) d = d.output_id();  // a job id names the job; what is read is its output proof
  return deps;
}

// plonky2 proofs one job costs in the reference: root jobs add minifier recursions (block_state_transition/mod.rs:125-126,
// root_aggregators/*/mod.rs:261,291, sighash_final_gl.rs:232-233), the sighash job is inner + 3 minifiers + wrapper
// (sighash_wrapper.rs:129-208) — SURVEY.md section 8(d): 46 jobs = 64 proofs for the example block
inline int proofs_per_job(uint8_t circuit_type) {
  switch (circuit_type) {
    case AggUserRegisterClaimDepositL2Transfer: case AggAddProcessL1WithdrawalAddL1Deposit: case GenerateRollupStateTransitionProof:
    case GenerateFinalSigHashProof: return 2;
    case GenerateSigHashIntrospectionProof: return 5;
    default: return 1;
  }
}

}  // namespace qb
