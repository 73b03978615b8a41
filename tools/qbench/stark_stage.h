// The SHA-256 STARK of a GenerateSigHashIntrospectionProof job, on a SYNTHETIC AIR of the reference's shape.
//
// In the reference the sighash circuit's witness generation proves a starkyx `ByteStark` first and verifies it inside the circuit
// (city_rollup_circuit/src/sighash_circuits/sighash.rs:132-146 -> city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524;
// 418 free + 912 extended columns :55-79; 2^k rows :310-312): three STARK proofs per block. The AIR lives in an absent crate, so with
// --stark-log-rows K the harness runs the generic prover (include/cityprover.h cp_stark_prove) on a stand-in of the same SHAPE: a
// seeded straight-line constraint program of >= 10^4 ops in the form of an instruction-list AIR (gadgets of a few columns, 10-40
// arithmetic ops, a few constraints of degree <= 3; every one of the 1 330 columns read; 2 alphas, 6 round challenges), the 912
// extended columns filled on the device by a map program + 304 cubic inversions + 912 prefix sums, rate 2, 84 queries, 16-bit
// proof of work — the configuration tools/bench_stark_air.py measures. The trace is random, so the proof is NOT a valid proof of
// anything; what the bytes of this prover are held against is tests/test_gpu_air.py / test_gpu_sha256_air.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "cityprover.h"

namespace qb {

struct StarkStage {
  static constexpr uint32_t K0 = 418, K1 = 912, N_PUBLIC = 4, N_CHALLENGE = 6;
  static constexpr uint64_t P = 0xFFFFFFFF00000001ull;
  int log_rows = 0;
  cp_air_program *cons = nullptr, *map = nullptr;
  cp_stark_step steps[3];
  cp_stark_desc desc;
  uint64_t *trace = nullptr;  // page-locked, K0 x n
  uint64_t publics[N_PUBLIC] = {3, 5, 7, 11};
  size_t n_ops = 0, n_constraints = 0;

  struct Gen {
    std::vector<cp_air_op> ops;
    std::vector<uint64_t> consts;
    std::vector<int> deg;  // degree of the value an op defines (-1: none)
    uint64_t x;
    explicit Gen(uint64_t seed) : x(seed) {}
    uint64_t next() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
    uint32_t emit(uint32_t op, uint32_t a, uint32_t b, int d) {
      ops.push_back(cp_air_op{op, a, b, 0});
      deg.push_back(d);
      return (uint32_t)ops.size() - 1;
    }
    uint32_t constant(uint64_t v) {
      consts.push_back(v % P);
      return emit(CP_AIR_CONST, (uint32_t)consts.size() - 1, 0, 0);
    }
  };

  // gadgets until every column has been read and the program has `want_ops` ops
  static void constraint_program(Gen &g, uint32_t n_columns, size_t want_ops, size_t *n_constraints) {
    uint32_t col = 0;
    size_t sinks = 0;
    while (g.ops.size() < want_ops || col < n_columns) {
      std::vector<uint32_t> vals;
      const int width = 4 + (int)(g.next() % 9);
      for (int i = 0; i < width; i++) {
        const uint32_t c = col < n_columns ? col++ : (uint32_t)(g.next() % n_columns);
        vals.push_back(g.emit(g.next() % 4 == 0 ? CP_AIR_NEXT : CP_AIR_LOCAL, c, 0, 1));
      }
      vals.push_back(g.emit(CP_AIR_CHALLENGE, (uint32_t)(g.next() % N_CHALLENGE), 0, 0));
      vals.push_back(g.next() % 2 ? g.emit(CP_AIR_PUBLIC, (uint32_t)(g.next() % N_PUBLIC), 0, 0) : g.constant(g.next()));
      const int n_arith = 10 + (int)(g.next() % 31);
      for (int i = 0; i < n_arith; i++) {
        const uint32_t a = vals[g.next() % vals.size()], b = vals[g.next() % vals.size()];
        const int da = g.deg[a], db = g.deg[b];
        const uint64_t r = g.next() % 8;
        if (r < 4 && da + db <= 3) vals.push_back(g.emit(CP_AIR_MUL, a, b, da + db));
        else if (r < 6) vals.push_back(g.emit(CP_AIR_ADD, a, b, da > db ? da : db));
        else if (r < 7) vals.push_back(g.emit(CP_AIR_SUB, a, b, da > db ? da : db));
        else vals.push_back(g.emit(CP_AIR_NEG, a, 0, da));
      }
      const int n_sinks = 3 + (int)(g.next() % 6);
      for (int i = 0; i < n_sinks; i++) {
        uint32_t v = vals[vals.size() - 1 - (g.next() % (vals.size() / 2))];
        if (g.deg[v] < 1) v = vals[0];
        const uint64_t r = g.next() % 16;
        const uint32_t kind = r < 9 ? CP_AIR_ASSERT_ZERO : r < 14 ? CP_AIR_ASSERT_ZERO_TRANSITION : r < 15 ? CP_AIR_ASSERT_ZERO_FIRST_ROW : CP_AIR_ASSERT_ZERO_LAST_ROW;
        // first / last row constraints count one degree more (their Lagrange factor): keep those at degree <= 2
        if (kind >= CP_AIR_ASSERT_ZERO_FIRST_ROW && g.deg[v] > 2) v = vals[0];
        g.emit(kind, v, 0, -1);
        sinks++;
      }
    }
    *n_constraints = sinks;
  }
  // extended column j = local(j mod K0) * challenge(j mod 6) + next((7 j + 1) mod K0) [- public(j mod 4)]
  static void map_program(Gen &g) {
    uint32_t ch[N_CHALLENGE];
    for (uint32_t i = 0; i < N_CHALLENGE; i++) ch[i] = g.emit(CP_AIR_CHALLENGE, i, 0, 0);
    for (uint32_t j = 0; j < K1; j++) {
      const uint32_t l = g.emit(CP_AIR_LOCAL, j % K0, 0, 1), nx = g.emit(CP_AIR_NEXT, (7 * j + 1) % K0, 0, 1);
      uint32_t v = g.emit(CP_AIR_ADD, g.emit(CP_AIR_MUL, l, ch[j % N_CHALLENGE], 1), nx, 1);
      if (j % 3) v = g.emit(CP_AIR_SUB, v, g.emit(CP_AIR_PUBLIC, j % N_PUBLIC, 0, 0), 1);
      g.emit(CP_AIR_STORE, j, v, -1);
    }
  }

  void open(cp_ctx *ctx, int log_rows_) {
    log_rows = log_rows_;
    const size_t n = (size_t)1 << log_rows;
    auto check = [&](int rc, const char *what) { if (rc != CP_OK) throw std::runtime_error(std::string(what) + ": " + cp_last_error(ctx)); };
    Gen gc(0x5A17A5EEDull), gm(1);
    constraint_program(gc, K0 + K1, 10500, &n_constraints);
    n_ops = gc.ops.size();
    map_program(gm);
    cp_air_program_desc dc;
    memset(&dc, 0, sizeof dc);
    dc.kind = CP_AIR_CONSTRAINTS; dc.ops = gc.ops.data(); dc.n_ops = gc.ops.size(); dc.consts = gc.consts.data(); dc.n_consts = gc.consts.size();
    dc.n_columns = K0 + K1; dc.n_public = N_PUBLIC; dc.n_challenge = N_CHALLENGE;
    cons = cp_air_program_create(ctx, &dc);
    if (!cons) throw std::runtime_error(std::string("cp_air_program_create (constraints): ") + cp_last_error(ctx));
    cp_air_program_desc dm = dc;
    dm.kind = CP_AIR_MAP; dm.ops = gm.ops.data(); dm.n_ops = gm.ops.size(); dm.consts = gm.consts.data(); dm.n_consts = gm.consts.size();
    dm.n_out_columns = K1;
    map = cp_air_program_create(ctx, &dm);
    if (!map) throw std::runtime_error(std::string("cp_air_program_create (map): ") + cp_last_error(ctx));
    memset(steps, 0, sizeof steps);
    steps[0].kind = CP_STARK_STEP_MAP; steps[0].program = map;
    steps[1].kind = CP_STARK_STEP_CUBIC_INVERSE; steps[1].first = 0; steps[1].count = K1 / 3; steps[1].modulus[0] = P - 1; steps[1].modulus[1] = 1;  // X^3 = X - 1
    steps[2].kind = CP_STARK_STEP_PREFIX_SUM; steps[2].first = 0; steps[2].count = K1;
    memset(&desc, 0, sizeof desc);
    desc.degree_bits = log_rows; desc.quotient_degree_bits = 1; desc.num_challenges = 2;
    desc.fri.degree_bits = log_rows; desc.fri.rate_bits = 1; desc.fri.cap_height = 4; desc.fri.pow_bits = 16; desc.fri.num_query_rounds = 84;
    int db = log_rows, na = 0;   // FriReductionStrategy::ConstantArityBits(4, 5)
    while (db > 5 && db + 1 - 4 >= 4) { desc.fri.arity_bits[na++] = 4; db -= 4; }
    desc.fri.n_arity = na;
    desc.n_trace_columns = K0; desc.n_extended_columns = K1; desc.n_round_challenges = N_CHALLENGE; desc.n_public = N_PUBLIC;
    desc.steps = steps; desc.n_steps = 3; desc.constraints = cons;
    void *p = nullptr;
    check(cp_host_alloc(ctx, (size_t)K0 * n * 8, &p), "cp_host_alloc");
    trace = (uint64_t *)p;
    uint64_t x = 0x243F6A8885A308D3ull;
    for (size_t i = 0; i < (size_t)K0 * n; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      trace[i] = x % P;
    }
  }
  // one proof; returns its length in bytes
  size_t prove(cp_ctx *ctx, uint64_t seed) {
    cp_challenger_state ch;
    memset(&ch, 0, sizeof ch);
    const uint64_t first[2] = {seed % P, 0x57A4Cull};
    if (cp_challenger_observe(&ch, first, 2) != CP_OK) throw std::runtime_error(std::string("cp_challenger_observe: ") + cp_last_error(nullptr));
    uint8_t *proof = nullptr;
    size_t len = 0;
    if (cp_stark_prove(ctx, &desc, trace, 0, publics, nullptr, &ch, 0, 0, &proof, &len) != CP_OK)
      throw std::runtime_error(std::string("cp_stark_prove: ") + cp_last_error(ctx));
    cp_free(proof);
    return len;
  }
  void close(cp_ctx *ctx) {
    if (trace) cp_host_free(ctx, trace);
    cp_air_program_destroy(cons);
    cp_air_program_destroy(map);
    trace = nullptr; cons = map = nullptr;
  }
};

}  // namespace qb
