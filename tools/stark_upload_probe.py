import sys, os, time, json, ctypes
sys.path.insert(0, "city-rollup_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np
import cityprover as cp
import bench_stark_air as B
import air_programs as A
from bench_stark_fri import arity_for
p = cp.Prover(0)
out = {}
for lg in (14, 16):
    n = 1 << lg
    rng = np.random.default_rng(1)
    t = rng.integers(0, cp.P, size=(B.K0, n), dtype=np.uint64)
    buf = cp.DeviceBuffer(p, t.size)
    ts = []
    for _ in range(4):
        p.sync(); t0 = time.perf_counter(); buf.upload(t); ts.append(time.perf_counter() - t0)
    pin = p.pinned(t)
    tp = []
    for _ in range(4):
        p.sync(); t0 = time.perf_counter(); p._check(p.lib.cp_h2d(p.ctx, buf.ptr, pin.ctypes.data, t.size * 8)); tp.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ok = bool((t < np.uint64(cp.P)).all()); tc = time.perf_counter() - t0
    cons_b, map_b = B.programs()
    cons, mp = cons_b.gpu(p), map_b.gpu(p)
    pub = rng.integers(0, cp.P, 4, dtype=np.uint64)
    desc, keep = cp.stark_desc(lg, 1, 2, cp.fri_params(lg, 1, 4, 16, 84, arity_for(lg, 1, 4)), B.K0, cons, B.K1, 6, n_public=4,
                               steps=[("map", mp), ("cubic_inverse", 0, B.K1 // 3, A.CUBIC_MODULUS), ("prefix_sum", 0, B.K1, False)])
    res = {}
    for name, arr in (("pageable", t), ("pinned", pin)):
        tt = []
        for _ in range(4):
            st = cp.ChallengerState(); p.sync(); t0 = time.perf_counter(); cp.stark_prove(p, desc, arr, st, publics=pub); tt.append(time.perf_counter() - t0)
        res[name] = sorted(tt)[1] * 1e3
    # device-resident trace
    tt = []
    for _ in range(4):
        st = cp.ChallengerState(); p.sync(); t0 = time.perf_counter()
        o, ln = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        pp = pub.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        p._check(p.lib.cp_stark_prove(p.ctx, ctypes.byref(desc), buf.ptr, 1, pp, None, ctypes.byref(st), 0, 0, ctypes.byref(o), ctypes.byref(ln)))
        tt.append(time.perf_counter() - t0); p.lib.cp_free(o)
    res["device"] = sorted(tt)[1] * 1e3
    out[lg] = {"MB": t.size * 8 / 1e6, "h2d_pageable_ms": sorted(ts)[1] * 1e3, "h2d_pinned_ms": sorted(tp)[1] * 1e3, "numpy_canonical_check_ms": tc * 1e3, "stark_prove_ms": res}
    cons.close(); mp.close(); buf.free()
print(json.dumps(out, indent=1))
