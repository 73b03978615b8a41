// A circuit pack: the circuits (".cpcirc", include/cityprover.h "circuit files") and witnesses (".cpwit") the harness binds
// to the job types of a dump. In the reference this binding is `CRWorkerToolboxRootCircuits::new`
// (city_rollup_circuit/src/worker/toolbox/root.rs:75-139, dispatch at :229-253 and toolbox/circuits.rs:414-486): one
// built circuit per job type, some jobs proving several circuits in a row. Here the circuits arrive as files — dumped
// from the real worker by the Rust side of the bridge (rust/plonky2-hwa-patch), or the shape-equivalent synthetic ones
// of tools/make_circuit_pack.py — and witness generation (A2) is outside the build, so each
// (job type, stage) comes with the wire matrix to prove.
//
// pack.manifest, one binding per line ('#' starts a comment):
//     <circuit_type | default> <stage> <circuit file> <witness file> [<witness file> ...]
// stage s of a job of that type proves <circuit file>; the k-th job of that type in a block (jobs ordered by their 24-byte
// id) proves it on the (k mod number of witnesses)-th witness file — SURVEY.md section 8(d) M1: one witness per job, "seed =
// job index". `default` binds every type not listed.
//
// .cpwit (cityprover/files.py writes the same): "CPWITNv1" | u32 version = 1 | u32 flags (bit 0: proof bytes present) |
// u64 circuit_digest[4] | u32 num_wires, u32 degree_bits, u32 n_public_inputs, u32 0 | public inputs | wires
// [num_wires][n] u64 | (u64 proof_len, proof bytes, zero padding to 8) | u64 FNV-1a 64 of everything before.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "jobs.h"

namespace qb {

struct Witness {
  uint64_t digest[4];
  uint32_t num_wires = 0, degree_bits = 0;
  std::vector<uint64_t> public_inputs, wires;
  std::vector<uint8_t> expected_proof;  // empty: none recorded
};

inline uint64_t fnv1a64(const uint8_t *p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

inline Witness load_witness(const std::string &path) {
  const std::vector<uint8_t> b = read_file(path);
  if (b.size() < 72 || memcmp(b.data(), "CPWITNv1", 8) != 0) throw ParseError(path + ": not a witness file");
  uint32_t version, flags, hdr[4];
  memcpy(&version, &b[8], 4);
  memcpy(&flags, &b[12], 4);
  if (version != 1) throw ParseError(path + ": witness file version " + std::to_string(version));
  uint64_t sum;
  memcpy(&sum, &b[b.size() - 8], 8);
  if (fnv1a64(b.data(), b.size() - 8) != sum) throw ParseError(path + ": checksum mismatch");
  Witness w;
  memcpy(w.digest, &b[16], 32);
  memcpy(hdr, &b[48], 16);
  w.num_wires = hdr[0];
  w.degree_bits = hdr[1];
  if (w.degree_bits > 26 || w.num_wires > 4096 || hdr[2] > (1u << 20)) throw ParseError(path + ": header out of range");
  size_t o = 64;
  const size_t n_pi = hdr[2], n_w = (size_t)w.num_wires << w.degree_bits;
  if (b.size() < o + 8 * (n_pi + n_w) + 8) throw ParseError(path + ": truncated");
  w.public_inputs.resize(n_pi);
  memcpy(w.public_inputs.data(), &b[o], 8 * n_pi);
  o += 8 * n_pi;
  w.wires.resize(n_w);
  memcpy(w.wires.data(), &b[o], 8 * n_w);
  o += 8 * n_w;
  if (flags & 1) {
    uint64_t len;
    if (b.size() < o + 16) throw ParseError(path + ": truncated proof");
    memcpy(&len, &b[o], 8);
    if (len > b.size() - o - 16) throw ParseError(path + ": truncated proof");
    w.expected_proof.assign(b.begin() + o + 8, b.begin() + o + 8 + len);
  }
  return w;
}

struct Binding {
  int circuit = -1;             // index into Pack::circuit_files
  std::vector<int> witnesses;   // indices into Pack::witnesses, one per job of this type in a block (at least one)
  int witness_for(size_t job_ordinal) const { return witnesses[job_ordinal % witnesses.size()]; }
};

struct Pack {
  std::string dir;
  std::vector<std::string> circuit_files;                 // distinct circuit files (absolute paths)
  std::vector<std::shared_ptr<Witness>> witnesses;         // distinct witness files
  std::vector<int> witness_circuit;                        // witness -> the circuit it belongs to
  std::map<int, std::vector<Binding>> by_type;             // circuit_type (-1 = default) -> stages

  const std::vector<Binding> &stages_for(uint8_t circuit_type) const {
    auto it = by_type.find(circuit_type);
    if (it == by_type.end()) it = by_type.find(-1);
    if (it == by_type.end()) throw ParseError("the circuit pack has no binding for circuit type " + std::to_string(circuit_type) + " and no default");
    return it->second;
  }
};

inline Pack load_pack(const std::string &dir, bool with_witnesses = true) {
  Pack p;
  p.dir = dir;
  const std::vector<uint8_t> raw = read_file(dir + "/pack.manifest");
  std::istringstream in(std::string(raw.begin(), raw.end()));
  std::map<std::string, int> circ_idx, wit_idx;
  std::string line;
  int lineno = 0;
  while (std::getline(in, line)) {
    lineno++;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.resize(hash);
    std::istringstream ls(line);
    std::string type, cfile, wfile;
    int stage;
    if (!(ls >> type)) continue;
    if (!(ls >> stage >> cfile >> wfile) || stage < 0 || stage > 15) throw ParseError("pack.manifest:" + std::to_string(lineno) + ": expected <type> <stage> <circuit> <witness>...");
    int t = -1;
    if (type != "default") {
      char *end = nullptr;
      const long v = strtol(type.c_str(), &end, 10);
      if (!end || *end || v < 0 || v > 255) throw ParseError("pack.manifest:" + std::to_string(lineno) + ": bad circuit type " + type);
      t = (int)v;
    }
    if (!circ_idx.count(cfile)) { circ_idx[cfile] = (int)p.circuit_files.size(); p.circuit_files.push_back(dir + "/" + cfile); }
    Binding b;
    b.circuit = circ_idx[cfile];
    do {
      if (!wit_idx.count(wfile)) {
        wit_idx[wfile] = (int)p.witnesses.size();
        p.witnesses.push_back(with_witnesses ? std::make_shared<Witness>(load_witness(dir + "/" + wfile)) : nullptr);
        p.witness_circuit.push_back(circ_idx[cfile]);
      } else if (p.witness_circuit[wit_idx[wfile]] != circ_idx[cfile]) {
        throw ParseError("pack.manifest:" + std::to_string(lineno) + ": witness " + wfile + " is bound to two circuits");
      }
      b.witnesses.push_back(wit_idx[wfile]);
    } while (ls >> wfile);
    auto &st = p.by_type[t];
    if ((int)st.size() != stage) throw ParseError("pack.manifest:" + std::to_string(lineno) + ": stages of a type must be listed in order from 0");
    st.push_back(b);
  }
  if (p.by_type.empty()) throw ParseError("pack.manifest binds nothing");
  return p;
}

}  // namespace qb
