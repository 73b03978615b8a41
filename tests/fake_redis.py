"""A small in-process Redis for the tests of `cityprover_qbench --mode redis-worker`: RESP2 over TCP, binary-safe, the
dozen commands the reference's worker issues through redis-rs and rsmq_async (city_redis_store/src/lib.rs:53-112,
city_rollup_worker_dispatch/src/implementations/redis/mod.rs:60-149), one lock around every command so several workers can
hammer it. EVAL knows one script: rsmq's popMessage (recognised by its ZRANGEBYSCORE ... LIMIT 0 1 head), run natively.
TIME is a logical clock (+1 ms per call): message order is then deterministic, as on a quiet real server."""
import socketserver
import threading


class Store:
    def __init__(self):
        self.lock = threading.RLock()
        self.hashes = {}     # key -> {field(bytes): value(bytes)}
        self.zsets = {}      # key -> {member(bytes): score(int)}
        self.clock_us = 1_700_000_000_000_000
        self.log = []        # (command name, args) of every command run, for the assertions
        self.popped = {}     # queue key -> [bodies in pop order]

    # ---- what the tests use directly ----
    def create_queue(self, name, vt=30, delay=0, maxsize=-1):
        """rsmq createQueue: the attributes hash rsmq:<name>:Q and the member of rsmq:QUEUES"""
        with self.lock:
            self.hashes.setdefault(b"rsmq:" + name.encode() + b":Q", {}).update(
                {b"vt": str(vt).encode(), b"delay": str(delay).encode(), b"maxsize": str(maxsize).encode(), b"totalrecv": b"0", b"totalsent": b"0"})

    def send(self, name, body):
        with self.lock:
            key = b"rsmq:" + name.encode()
            self.clock_us += 1000
            mid = ("%012x" % self.clock_us).encode() + b"-test"
            self.zsets.setdefault(key, {})[mid] = self.clock_us // 1000
            self.hashes[key + b":Q"][mid] = body if isinstance(body, bytes) else body.encode()

    def queue_bodies(self, name):
        with self.lock:
            key = b"rsmq:" + name.encode()
            z = self.zsets.get(key, {})
            return [self.hashes[key + b":Q"][m] for m in sorted(z, key=lambda m: (z[m], m))]

    # ---- commands ----
    def run(self, args):
        name = args[0].upper().decode()
        with self.lock:
            self.log.append((name, args[1:]))
            return getattr(self, "cmd_" + name)(*args[1:])

    def cmd_HGET(self, key, field):
        return self.hashes.get(key, {}).get(field)

    def cmd_HSETNX(self, key, field, value):
        h = self.hashes.setdefault(key, {})
        if field in h:
            return 0
        h[field] = value
        return 1

    def cmd_HSET(self, key, *fv):
        h = self.hashes.setdefault(key, {})
        new = 0
        for f, v in zip(fv[::2], fv[1::2]):
            new += f not in h
            h[f] = v
        return new

    def cmd_HINCRBY(self, key, field, by):
        h = self.hashes.setdefault(key, {})
        v = int(h.get(field, b"0")) + int(by)
        h[field] = str(v).encode()
        return v

    def cmd_HMGET(self, key, *fields):
        h = self.hashes.get(key, {})
        return [h.get(f) for f in fields]

    def cmd_HDEL(self, key, *fields):
        h = self.hashes.get(key, {})
        return sum(h.pop(f, None) is not None for f in fields)

    def cmd_ZADD(self, key, score, member):
        z = self.zsets.setdefault(key, {})
        new = member not in z
        z[member] = int(score)
        return int(new)

    def cmd_ZCARD(self, key):
        return len(self.zsets.get(key, {}))

    def cmd_TIME(self):
        self.clock_us += 1000
        return [str(self.clock_us // 1_000_000).encode(), str(self.clock_us % 1_000_000).encode()]

    def cmd_PING(self):
        return "PONG"

    def cmd_EVAL(self, script, nkeys, *keys):
        if b"ZRANGEBYSCORE" not in script or int(nkeys) != 2:
            raise ValueError("unknown script")
        key, now = keys[0], int(keys[1])
        z = self.zsets.get(key, {})
        visible = sorted((s, m) for m, s in z.items() if s <= now)
        if not visible:
            return []
        mid = visible[0][1]
        q = self.hashes[key + b":Q"]
        q[b"totalrecv"] = str(int(q.get(b"totalrecv", b"0")) + 1).encode()
        body = q[mid]
        del z[mid]
        for suffix in (b"", b":rc", b":fr"):
            q.pop(mid + suffix, None)
        self.popped.setdefault(key, []).append(body)
        return [mid, body, 1, keys[1]]


def encode(v):
    if v is None:
        return b"$-1\r\n"
    if isinstance(v, bool) or isinstance(v, int):
        return b":%d\r\n" % int(v)
    if isinstance(v, str):
        return b"+" + v.encode() + b"\r\n"
    if isinstance(v, bytes):
        return b"$%d\r\n" % len(v) + v + b"\r\n"
    return b"*%d\r\n" % len(v) + b"".join(encode(x) for x in v)


class Handler(socketserver.StreamRequestHandler):
    def read_command(self):
        line = self.rfile.readline()
        if not line:
            return None
        assert line[:1] == b"*", line
        args = []
        for _ in range(int(line[1:])):
            n = int(self.rfile.readline()[1:])
            args.append(self.rfile.read(n))
            self.rfile.read(2)
        return args

    def handle(self):
        store = self.server.store
        queued = None
        while True:
            try:
                args = self.read_command()
            except (ConnectionError, ValueError):
                return
            if args is None:
                return
            name = args[0].upper()
            try:
                if name == b"MULTI":
                    queued, out = [], "OK"
                elif name == b"EXEC":
                    with store.lock:
                        out = [store.run(a) for a in queued]
                    queued = None
                elif queued is not None:
                    queued.append(args)
                    out = "QUEUED"
                else:
                    out = store.run(args)
                self.wfile.write(encode(out))
            except Exception as e:  # noqa: BLE001 - a Redis error reply, whatever went wrong
                self.wfile.write(b"-ERR " + str(e).encode() + b"\r\n")
            self.wfile.flush()


class FakeRedis:
    """with FakeRedis() as r: r.uri is "127.0.0.1:<port>", r.store the data"""

    def __enter__(self):
        socketserver.ThreadingTCPServer.allow_reuse_address = True
        self.server = socketserver.ThreadingTCPServer(("127.0.0.1", 0), Handler)
        self.server.daemon_threads = True
        self.server.store = self.store = Store()
        self.uri = "127.0.0.1:%d" % self.server.server_address[1]
        self.thread = threading.Thread(target=self.server.serve_forever, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.server.shutdown()
        self.server.server_close()
