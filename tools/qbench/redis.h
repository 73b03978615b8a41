// The Redis half of the job runner: what a GPU worker needs to join a LIVE city-rollup
// deployment instead of replaying a dump — the production proof store and the job queue, spoken natively:
//
//   RedisStore   city_redis_store/src/lib.rs:53-112        proofs and witnesses in the hash `proofs` (HGET / HSETNX, the key is the
//                                                          24-byte QProvingJobDataID), group counters in `proof_counters` (HINCRBY)
//   RedisQueue   city_rollup_worker_dispatch/src/implementations/redis/mod.rs:47-150   RSMQ queues JOB / NOTIFICATIONS:
//                pop_one = rsmq `pop_message` (destructive), dispatch = `send_message` with a serde_json payload
//   the loop     city_rollup_core_worker/src/event_processor.rs:29-62, src/lib.rs:131-145
//
// RSMQ's wire form (the `rsmq` crate, a port of the Node library; UPSTREAM-MEMORY, not in the tree): per queue a sorted set
// `rsmq:<q>` (member = message id, score = the time in ms at which it becomes visible) and a hash `rsmq:<q>:Q` (field <id> =
// the body, <id>:rc / <id>:fr = receive count / first receive, plus the queue attributes vt, delay, maxsize, totalsent,
// totalrecv). pop_message is one Lua script (below) run with EVAL; send_message is TIME + HMGET of the attributes + a
// MULTI block. Only what the worker loop touches is implemented. Plain POSIX sockets, RESP2; no TLS, no AUTH.
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "jobs.h"

namespace qb {

struct RedisError : std::runtime_error { using std::runtime_error::runtime_error; };

struct Reply {
  enum Kind { Nil, Status, Error, Integer, Bulk, Array } kind = Nil;
  std::string str;        // Status / Error / Bulk
  long long integer = 0;  // Integer
  std::vector<Reply> items;
};

class RespClient {
  int fd = -1;
  std::string buf;
  size_t pos = 0;

  void fill() {
    if (pos > 0 && pos == buf.size()) { buf.clear(); pos = 0; }
    char tmp[65536];
    const ssize_t n = ::recv(fd, tmp, sizeof tmp, 0);
    if (n <= 0) throw RedisError("redis: connection closed");
    buf.append(tmp, (size_t)n);
  }
  std::string line() {
    for (;;) {
      const size_t e = buf.find("\r\n", pos);
      if (e != std::string::npos) {
        std::string l = buf.substr(pos, e - pos);
        pos = e + 2;
        return l;
      }
      fill();
    }
  }
  Reply parse() {
    const std::string l = line();
    if (l.empty()) throw RedisError("redis: empty reply line");
    Reply r;
    switch (l[0]) {
      case '+': r.kind = Reply::Status; r.str = l.substr(1); return r;
      case '-': r.kind = Reply::Error; r.str = l.substr(1); return r;
      case ':': r.kind = Reply::Integer; r.integer = atoll(l.c_str() + 1); return r;
      case '$': {
        const long long n = atoll(l.c_str() + 1);
        if (n < 0) return r;  // nil
        if (n > (1ll << 30)) throw RedisError("redis: bulk string too large");
        while (buf.size() - pos < (size_t)n + 2) fill();
        r.kind = Reply::Bulk;
        r.str = buf.substr(pos, (size_t)n);
        pos += (size_t)n + 2;
        return r;
      }
      case '*': {
        const long long n = atoll(l.c_str() + 1);
        if (n < 0) return r;
        if (n > (1 << 24)) throw RedisError("redis: array too large");
        r.kind = Reply::Array;
        for (long long i = 0; i < n; i++) r.items.push_back(parse());
        return r;
      }
      default: throw RedisError("redis: unexpected reply type '" + l.substr(0, 1) + "'");
    }
  }

 public:
  RespClient() = default;
  RespClient(const RespClient &) = delete;
  ~RespClient() { if (fd >= 0) ::close(fd); }
  // "host:port" (redis://host:port/ is accepted too)
  void connect(std::string uri) {
    if (uri.rfind("redis://", 0) == 0) uri = uri.substr(8);
    while (!uri.empty() && uri.back() == '/') uri.pop_back();
    const size_t c = uri.rfind(':');
    const std::string host = c == std::string::npos ? uri : uri.substr(0, c), port = c == std::string::npos ? "6379" : uri.substr(c + 1);
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) != 0 || !res) throw RedisError("redis: cannot resolve " + uri);
    for (addrinfo *a = res; a; a = a->ai_next) {
      fd = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
      if (fd < 0) continue;
      if (::connect(fd, a->ai_addr, a->ai_addrlen) == 0) break;
      ::close(fd);
      fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) throw RedisError("redis: cannot connect to " + uri);
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  }
  // one command (binary-safe arguments) -> its reply; a Redis error reply becomes an exception
  Reply command(const std::vector<std::string> &args) {
    std::string out = "*" + std::to_string(args.size()) + "\r\n";
    for (const auto &a : args) {
      out += "$" + std::to_string(a.size()) + "\r\n";
      out += a;
      out += "\r\n";
    }
    size_t sent = 0;
    while (sent < out.size()) {
      const ssize_t n = ::send(fd, out.data() + sent, out.size() - sent, MSG_NOSIGNAL);
      if (n <= 0) throw RedisError("redis: send failed");
      sent += (size_t)n;
    }
    Reply r = parse();
    if (r.kind == Reply::Error) throw RedisError("redis: " + r.str);
    return r;
  }
};

inline std::string key24(const JobId &id) {
  const auto b = id.bytes();
  return std::string((const char *)b.data(), 24);
}

// QProofStoreReaderSync / WriterSync over Redis, the subset process_job uses (city_redis_store/src/lib.rs:53-112)
class RedisStore {
  RespClient &c;

 public:
  explicit RedisStore(RespClient &client) : c(client) {}
  std::vector<uint8_t> get_bytes(const JobId &id) {  // get_bytes_by_id: HGET proofs <id>
    const Reply r = c.command({"HGET", "proofs", key24(id)});
    if (r.kind != Reply::Bulk) throw StoreError("Data not found. Wanted " + id.hex());
    return std::vector<uint8_t>(r.str.begin(), r.str.end());
  }
  void set_bytes(const JobId &id, const std::vector<uint8_t> &v) {  // set_bytes_by_id: HSETNX (first writer wins: idempotent re-proving)
    c.command({"HSETNX", "proofs", key24(id), std::string(v.begin(), v.end())});
  }
  uint32_t inc_counter(const JobId &id) {  // inc_counter_by_id: HINCRBY proof_counters <id> 1
    const Reply r = c.command({"HINCRBY", "proof_counters", key24(id), "1"});
    if (r.kind != Reply::Integer) throw StoreError("HINCRBY did not return an integer");
    return (uint32_t)r.integer;
  }
  uint32_t get_goal(const JobId &job) {  // proof_store.rs:15-21
    const auto g = get_bytes(job.goal_id_of_counter());
    if (g.size() != 4) throw StoreError("goal record of " + job.hex() + " is not 4 bytes");
    uint32_t v;
    memcpy(&v, g.data(), 4);
    return v;
  }
  std::vector<JobId> get_next_jobs(const JobId &job) {  // proof_store.rs:22-31
    const auto b = get_bytes(job.next_jobs_id_of_counter());
    uint64_t n = 0;
    if (b.size() < 8) throw StoreError("next-jobs record of " + job.hex() + " is truncated");
    memcpy(&n, b.data(), 8);
    // bound n by the record before multiplying: a huge count must not wrap 8 + 24 n round to the record's size
    if (n > (b.size() - 8) / 24 || b.size() != 8 + 24 * (size_t)n) throw StoreError("next-jobs record of " + job.hex() + " has the wrong length");
    std::vector<JobId> out(n);
    for (uint64_t i = 0; i < n; i++) out[i] = JobId::from_bytes(b.data() + 8 + 24 * i);
    return out;
  }
};

// serde_json of QProvingJobDataID (job_id.rs:205-215: derived Serialize, the enums through serde_repr = plain numbers)
inline std::string job_to_json(const JobId &j) {
  char b[256];
  snprintf(b, sizeof b, "{\"topic\":%u,\"goal_id\":%llu,\"circuit_type\":%u,\"group_id\":%u,\"sub_group_id\":%u,\"task_index\":%u,\"data_type\":%u,\"data_index\":%u}",
           (unsigned)j.topic, (unsigned long long)j.goal_id, (unsigned)j.circuit_type, j.group_id, j.sub_group_id, j.task_index, (unsigned)j.data_type,
           (unsigned)j.data_index);
  return b;
}
inline JobId job_from_json(const std::string &s) {
  auto num = [&](const char *key) -> unsigned long long {
    const std::string k = std::string("\"") + key + "\"";
    size_t p = s.find(k);
    if (p == std::string::npos) throw ParseError(std::string("job message without \"") + key + "\": " + s);
    p = s.find(':', p + k.size());
    if (p == std::string::npos) throw ParseError("malformed job message: " + s);
    p++;
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) p++;
    if (p >= s.size() || s[p] < '0' || s[p] > '9') throw ParseError(std::string("job message: \"") + key + "\" is not a number: " + s);
    return strtoull(s.c_str() + p, nullptr, 10);
  };
  JobId j;
  j.topic = (uint8_t)num("topic");
  j.goal_id = num("goal_id");
  j.circuit_type = (uint8_t)num("circuit_type");
  j.group_id = (uint32_t)num("group_id");
  j.sub_group_id = (uint32_t)num("sub_group_id");
  j.task_index = (uint32_t)num("task_index");
  j.data_type = (uint8_t)num("data_type");
  j.data_index = (uint8_t)num("data_index");
  if (!valid_enums(j)) throw ParseError("job message with an unknown topic / circuit type / data type: " + s);
  return j;
}

// RSMQ's popMessage script (rsmq: `popMessage`), KEYS[1] = rsmq:<q>, KEYS[2] = now (ms)
static const char *const RSMQ_POP_LUA =
    "local msg = redis.call(\"ZRANGEBYSCORE\", KEYS[1], \"-inf\", KEYS[2], \"LIMIT\", \"0\", \"1\")\n"
    "if #msg == 0 then return {} end\n"
    "redis.call(\"HINCRBY\", KEYS[1] .. \":Q\", \"totalrecv\", 1)\n"
    "local mbody = redis.call(\"HGET\", KEYS[1] .. \":Q\", msg[1])\n"
    "local rc = redis.call(\"HINCRBY\", KEYS[1] .. \":Q\", msg[1] .. \":rc\", 1)\n"
    "local o = {msg[1], mbody, rc}\n"
    "if rc==1 then table.insert(o, KEYS[2]) else local fr = redis.call(\"HGET\", KEYS[1] .. \":Q\", msg[1] .. \":fr\") table.insert(o, fr) end\n"
    "redis.call(\"ZREM\", KEYS[1], msg[1])\n"
    "redis.call(\"HDEL\", KEYS[1] .. \":Q\", msg[1], msg[1] .. \":rc\", msg[1] .. \":fr\")\n"
    "return o";

class RsmqQueue {
  RespClient &c;
  std::string ns;
  std::mt19937_64 rng{std::random_device{}()};

  // Redis TIME -> (ms, the microsecond remainder padded to six digits as rsmq uses it in message ids)
  std::pair<unsigned long long, std::string> now() {
    const Reply t = c.command({"TIME"});
    if (t.kind != Reply::Array || t.items.size() != 2) throw RedisError("redis: TIME");
    const unsigned long long s = strtoull(t.items[0].str.c_str(), nullptr, 10), us = strtoull(t.items[1].str.c_str(), nullptr, 10);
    char pad[16];
    snprintf(pad, sizeof pad, "%06llu", us);
    return {s * 1000 + us / 1000, std::to_string(s) + pad};
  }

 public:
  explicit RsmqQueue(RespClient &client, std::string name_space = "rsmq") : c(client), ns(std::move(name_space)) {}

  // pop_message: the oldest visible message, removed; false when the queue holds none
  bool pop(const std::string &q, std::string &body) {
    const auto t = now();
    const Reply r = c.command({"EVAL", RSMQ_POP_LUA, "2", ns + ":" + q, std::to_string(t.first)});
    if (r.kind != Reply::Array || r.items.size() < 2) return false;
    body = r.items[1].str;
    return true;
  }
  // send_message with the queue's own delay; the id is base36(seconds + microseconds) followed by 22 random characters
  void send(const std::string &q, const std::string &body) {
    const std::string key = ns + ":" + q;
    const Reply at = c.command({"HMGET", key + ":Q", "vt", "delay", "maxsize"});
    if (at.kind != Reply::Array || at.items.size() != 3 || at.items[0].kind != Reply::Bulk) throw RedisError("rsmq: queue " + q + " not found");
    const auto t = now();
    const unsigned long long delay_ms = strtoull(at.items[1].str.c_str(), nullptr, 10) * 1000;
    const long long maxsize = atoll(at.items[2].str.c_str());
    if (maxsize >= 0 && (long long)body.size() > maxsize) throw RedisError("rsmq: message too long for queue " + q);
    static const char *d36 = "0123456789abcdefghijklmnopqrstuvwxyz", *d62 = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789";
    unsigned long long v = strtoull(t.second.c_str(), nullptr, 10);
    std::string id;
    while (v) { id.insert(id.begin(), d36[v % 36]); v /= 36; }
    for (int i = 0; i < 22; i++) id += d62[rng() % 62];
    c.command({"MULTI"});
    c.command({"ZADD", key, std::to_string(t.first + delay_ms), id});
    c.command({"HSET", key + ":Q", id, body});
    c.command({"HINCRBY", key + ":Q", "totalsent", "1"});
    // EXEC answers with one reply per queued command (or Nil when the transaction was discarded): a command that failed inside
    // it would otherwise leave a half-sent message unnoticed
    const Reply ex = c.command({"EXEC"});
    if (ex.kind != Reply::Array || ex.items.size() != 3) throw RedisError("rsmq: sending to " + q + ": the transaction was not executed");
    for (const Reply &it : ex.items)
      if (it.kind == Reply::Error) throw RedisError("rsmq: sending to " + q + ": " + it.str);
  }
  // is_empty (redis/mod.rs:144-149): the number of messages of the queue
  long long size(const std::string &q) {
    const Reply r = c.command({"ZCARD", ns + ":" + q});
    return r.kind == Reply::Integer ? r.integer : 0;
  }
};

}  // namespace qb
