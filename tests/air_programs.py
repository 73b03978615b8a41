"""Straight-line AIR programs for the tests of the generic AIR machinery (include/cityprover.h cp_air_*, oracle/stark_air.c):
a builder that RECORDS arithmetic the way the Rust recording parser does (rust/starkyx-patch/recording_parser.rs: every parser
call appends an op and returns its index), seeded random programs at the SHA-256 STARK's width (418 + 912 columns,
city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:55-79), and a toy AIR WITH a lookup argument (logUp over a cubic
extension) whose constraints are written ONCE over an abstract field and interpreted three ways: recorded into a program,
evaluated on rows in Python integers, evaluated at zeta over F_p^2. Test infrastructure only."""
import numpy as np

P = 0xFFFFFFFF00000001
(LOCAL, NEXT, PUBLIC, GLOBAL, CHALLENGE, CONST, ADD, SUB, MUL, NEG, INV, ASSERT_ZERO, ASSERT_ZERO_TRANSITION, ASSERT_ZERO_FIRST_ROW,
 ASSERT_ZERO_LAST_ROW, STORE) = range(16)
CONSTRAINTS, MAP = 0, 1
SINKS = {"all": ASSERT_ZERO, "transition": ASSERT_ZERO_TRANSITION, "first": ASSERT_ZERO_FIRST_ROW, "last": ASSERT_ZERO_LAST_ROW}


class Builder:
    """records ops; values are op indices"""

    def __init__(self, kind, n_columns, n_public=0, n_global=0, n_challenge=0, n_out_columns=0):
        self.kind, self.n_columns, self.n_public, self.n_global, self.n_challenge, self.n_out_columns = kind, n_columns, n_public, n_global, n_challenge, n_out_columns
        self.ops, self.consts, self._const_ix = [], [], {}

    def _emit(self, op, a=0, b=0):
        self.ops.append((op, a, b, 0))
        return len(self.ops) - 1

    def local(self, c): return self._emit(LOCAL, c)
    def next(self, c): return self._emit(NEXT, c)
    def public(self, i): return self._emit(PUBLIC, i)
    def glob(self, i): return self._emit(GLOBAL, i)
    def challenge(self, i): return self._emit(CHALLENGE, i)

    def const(self, v):
        v %= P
        if v not in self._const_ix:
            self._const_ix[v] = len(self.consts)
            self.consts.append(v)
        return self._emit(CONST, self._const_ix[v])

    def add(self, a, b): return self._emit(ADD, a, b)
    def sub(self, a, b): return self._emit(SUB, a, b)
    def mul(self, a, b): return self._emit(MUL, a, b)
    def neg(self, a): return self._emit(NEG, a)
    def inv(self, a): return self._emit(INV, a)
    def assert_zero(self, a, when="all"): return self._emit(SINKS[when], a)
    def store(self, col, a): return self._emit(STORE, col, a)

    def arrays(self):
        return np.array(self.ops, dtype=np.uint32).reshape(-1, 4), np.array(self.consts, dtype=np.uint64)

    def kwargs(self):
        return dict(n_columns=self.n_columns, n_public=self.n_public, n_global=self.n_global, n_challenge=self.n_challenge, n_out_columns=self.n_out_columns)

    def gpu(self, prover):
        import cityprover
        ops, consts = self.arrays()
        return cityprover.AirProgram(prover, self.kind, ops, consts, **self.kwargs())

    def oracle(self):
        import oracle_lib as O
        ops, consts = self.arrays()
        return O.AirProgram(self.kind, ops, consts, **self.kwargs())


def random_program(seed, n_columns, n_ops, n_public=3, n_global=2, n_challenge=3, max_degree=3, far=0.1):
    """a seeded constraint program of ~n_ops ops over n_columns columns, nearly all of it LIVE: chains of arithmetic on recent
    values (short-lived temporaries), operands from far back with probability `far` (long-lived ones), every column read, values
    nobody has consumed yet preferred as operands and as constraints, sinks of all four kinds within the degree bound (first /
    last row constraints one lower)."""
    rng = np.random.default_rng(seed)
    b = Builder(CONSTRAINTS, n_columns, n_public, n_global, n_challenge)
    deg, vals, unused = {}, [], []

    def push(i, d):
        deg[i] = d
        vals.append(i)
        unused.append(i)
        return i

    def load():
        c = int(rng.integers(0, n_columns)) if rng.random() < 0.5 else len(vals) % n_columns
        return push(b.local(c) if rng.random() < 0.6 else b.next(c), 1)

    def pick():
        if unused and rng.random() < 0.7:
            return unused.pop(max(0, len(unused) - 1 - int(rng.exponential(2))))
        if rng.random() < far:
            return vals[int(rng.integers(0, len(vals)))]
        x = vals[max(0, len(vals) - 1 - int(rng.exponential(6)))]
        if x in unused[-8:]:
            unused.remove(x)
        return x

    for _ in range(4):
        load()
    n_sinks = 0
    while len(b.ops) < n_ops:
        r = rng.random()
        if r < 0.27 or len(vals) < 4:
            load()
        elif r < 0.33:
            k = int(rng.integers(0, 4))
            if k == 0:
                push(b.const(int(rng.integers(0, P, dtype=np.uint64))), 0)
            elif k == 1 and n_public:
                push(b.public(int(rng.integers(0, n_public))), 0)
            elif k == 2 and n_global:
                push(b.glob(int(rng.integers(0, n_global))), 0)
            elif n_challenge:
                push(b.challenge(int(rng.integers(0, n_challenge))), 0)
        elif r < 0.88 and len(unused) < 24:
            x, y = pick(), pick()
            k = rng.random()
            if k < 0.45 and deg[x] + deg[y] <= max_degree:
                push(b.mul(x, y), deg[x] + deg[y])
            elif k < 0.70:
                push(b.add(x, y), max(deg[x], deg[y]))
            elif k < 0.95:
                push(b.sub(x, y), max(deg[x], deg[y]))
            else:
                unused.append(y)
                push(b.neg(x), deg[x])
        else:
            x = pick()
            when = ("all", "transition", "first", "last")[int(rng.integers(0, 4))]
            if when in ("first", "last") and deg[x] + 1 > max_degree:
                when = "all"
            b.assert_zero(x, when)
            n_sinks += 1
    for x in unused[-64:]:               # what is still unconsumed ends in constraints as well
        b.assert_zero(x, "all")
        n_sinks += 1
    if not n_sinks:
        b.assert_zero(vals[-1], "all")
    return b


def gadget_program(seed, n_columns, n_ops, n_public=3, n_global=2, n_challenge=3, max_degree=3, share=0.03):
    """a seeded constraint program shaped like an AIR that is a list of instructions (byte operations, limb arithmetic, lookup
    accumulations): GADGETS of 4-12 columns, 10-40 arithmetic ops on their own values and a few uniform ones, 3-8 constraints
    each; with probability `share` an operand comes from an earlier gadget (a shared subexpression). Expression depth stays
    bounded, every column is read, all four kinds of constraint occur."""
    rng = np.random.default_rng(seed)
    b = Builder(CONSTRAINTS, n_columns, n_public, n_global, n_challenge)
    deg, old = {}, []
    col = 0
    while len(b.ops) < n_ops:
        vals = []
        for _ in range(int(rng.integers(4, 13))):
            c = col % n_columns if rng.random() < 0.7 else int(rng.integers(0, n_columns))
            col += 1
            v = b.local(c) if rng.random() < 0.65 else b.next(c)
            deg[v] = 1
            vals.append(v)
        for _ in range(int(rng.integers(0, 3))):
            k = int(rng.integers(0, 4))
            if k == 0:
                v = b.const(int(rng.integers(0, 1 << 16)) if rng.random() < 0.7 else int(rng.integers(0, P, dtype=np.uint64)))
            elif k == 1 and n_public:
                v = b.public(int(rng.integers(0, n_public)))
            elif k == 2 and n_global:
                v = b.glob(int(rng.integers(0, n_global)))
            elif n_challenge:
                v = b.challenge(int(rng.integers(0, n_challenge)))
            else:
                continue
            deg[v] = 0
            vals.append(v)
        unused = []
        for _ in range(int(rng.integers(10, 41))):
            def pick():
                if old and rng.random() < share:
                    return old[int(rng.integers(max(0, len(old) - 400), len(old)))]
                if unused and rng.random() < 0.6:
                    return unused.pop(int(rng.integers(0, len(unused))))
                return vals[int(rng.integers(0, len(vals)))]
            x, y = pick(), pick()
            k = rng.random()
            if k < 0.4 and deg[x] + deg[y] <= max_degree:
                v, d = b.mul(x, y), deg[x] + deg[y]
            elif k < 0.7:
                v, d = b.add(x, y), max(deg[x], deg[y])
            elif k < 0.95:
                v, d = b.sub(x, y), max(deg[x], deg[y])
            else:
                v, d = b.neg(x), deg[x]
            deg[v] = d
            vals.append(v)
            unused.append(v)
        n_c = int(rng.integers(3, 9))
        for x in (unused[-n_c:] if len(unused) >= n_c else unused):
            when = ("all", "all", "transition", "transition", "first", "last")[int(rng.integers(0, 6))]
            if when in ("first", "last") and deg[x] + 1 > max_degree:
                when = "all"
            b.assert_zero(x, when)
        old.extend(v for v in vals if deg[v] > 0)
    return b


# ---- cubic extension F_p[X]/(X^3 - m1 X - m0) over an abstract base field -------------------------------------------------
class IntField:
    """Python integers mod p"""
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    const = staticmethod(lambda v: v % P)


class ExtField:
    """F_p^2 = F_p[X]/(X^2 - 7) as pairs of Python integers"""
    zero, one = (0, 0), (1, 0)
    add = staticmethod(lambda a, b: ((a[0] + b[0]) % P, (a[1] + b[1]) % P))
    sub = staticmethod(lambda a, b: ((a[0] - b[0]) % P, (a[1] - b[1]) % P))
    mul = staticmethod(lambda a, b: ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P))
    const = staticmethod(lambda v: (v % P, 0))


class RecField:
    """records into a Builder: elements are op indices"""

    def __init__(self, b):
        self.b = b
        self.zero, self.one = b.const(0), b.const(1)
        self.add, self.sub, self.mul, self.const = b.add, b.sub, b.mul, b.const


def cubic_mul(F, m, x, y):
    """(x0 + x1 X + x2 X^2)(y0 + y1 X + y2 X^2) with X^3 = m1 X + m0; m = (m0, m1) as elements of F"""
    m0, m1 = m
    t = [F.mul(x[i], y[j]) for i in range(3) for j in range(3)]
    d0, d1, d2 = t[0], F.add(t[1], t[3]), F.add(F.add(t[2], t[4]), t[6])
    d3, d4 = F.add(t[5], t[7]), t[8]
    # X^3 = m1 X + m0;  X^4 = m1 X^2 + m0 X
    c0 = F.add(d0, F.mul(m0, d3))
    c1 = F.add(F.add(d1, F.mul(m1, d3)), F.mul(m0, d4))
    c2 = F.add(d2, F.mul(m1, d4))
    return (c0, c1, c2)


def cubic_add(F, x, y): return tuple(F.add(a, b) for a, b in zip(x, y))
def cubic_sub(F, x, y): return tuple(F.sub(a, b) for a, b in zip(x, y))
def cubic_scale(F, x, s): return tuple(F.mul(a, s) for a in x)


# ---- the toy AIR with a lookup ---------------------------------------------------------------------------------------------
# execution trace (7 columns): a, b, s — Fibonacci with a running sum of a*b*b —, v0, v1 — two values per row, each a member of
# the table —, t — the table —, m — how often t is looked up. Challenge beta in the cubic extension (3 base challenges).
# extended columns (15): e0 = 1/(beta - v0), e1 = 1/(beta - v1), e2 = 1/(beta - t), row = e0 + e1 - m e2, D = the running sum of
# `row` over the rows above (D[0] = 0). Constraints (degree <= 3):
#   a' = b, b' = a + b, s' = s + a b b (transition); a = 0, b = 1, s = 0 (first row)
#   (beta - v0) e0 = 1, (beta - v1) e1 = 1, (beta - t) e2 = 1; row = e0 + e1 - m e2 (every row)
#   D' = D + row (transition), D = 0 (first row), D + row = 0 (last row: the log-derivative sums of values and table agree)
LOOKUP_K0, LOOKUP_K1 = 7, 15
CUBIC_MODULUS = (P - 1, 1)   # X^3 = X - 1, i.e. F_p[X]/(X^3 - X + 1) (the form starkyx's cubic extension takes: UPSTREAM-MEMORY)


def lookup_trace(n, cheat=None):
    a, b, s = [0] * n, [0] * n, [0] * n
    b[0] = 1
    for i in range(1, n):
        a[i] = b[i - 1]
        b[i] = (a[i - 1] + b[i - 1]) % P
        s[i] = (s[i - 1] + a[i - 1] * b[i - 1] * b[i - 1]) % P
    t = [(7 * i + 3) % P for i in range(n)]
    v0 = [t[(3 * i) % n] for i in range(n)]
    v1 = [t[(i * i + 1) % n] for i in range(n)]
    if cheat == "value":
        v1[n // 2] = 5   # not in the table
    m = [0] * n
    pos = {v: i for i, v in enumerate(t)}
    for v in v0 + v1:
        if v in pos:
            m[pos[v]] += 1
    if cheat == "fib":
        b[n // 3] = (b[n // 3] + 1) % P
    return np.array([a, b, s, v0, v1, t, m], dtype=np.uint64)


def lookup_constraints(F, loc, nxt, beta, emit):
    """every constraint over the field F; loc / nxt: the 22 columns of a row (trace then extended); emit(value, when)"""
    m = (F.const(CUBIC_MODULUS[0]), F.const(CUBIC_MODULUS[1]))
    a, b, s, v0, v1, t, mult = loc[:7]
    a2, b2, s2 = nxt[:3]
    e = [tuple(loc[7 + 3 * j + c] for c in range(3)) for j in range(5)]     # e0, e1, e2, row, D
    e_next = [tuple(nxt[7 + 3 * j + c] for c in range(3)) for j in range(5)]
    emit(F.sub(a2, b), "transition")
    emit(F.sub(b2, F.add(a, b)), "transition")
    emit(F.sub(s2, F.add(s, F.mul(F.mul(a, b), b))), "transition")
    emit(a, "first")
    emit(F.sub(b, F.one), "first")
    emit(s, "first")
    one3 = (F.one, F.zero, F.zero)
    for val, inv in ((v0, e[0]), (v1, e[1]), (t, e[2])):
        den = (F.sub(beta[0], val), beta[1], beta[2])
        for c in cubic_sub(F, cubic_mul(F, m, den, inv), one3):
            emit(c, "all")
    row_want = cubic_sub(F, cubic_add(F, e[0], e[1]), cubic_scale(F, e[2], mult))
    for c in cubic_sub(F, e[3], row_want):
        emit(c, "all")
    for c in cubic_sub(F, e_next[4], cubic_add(F, e[4], e[3])):
        emit(c, "transition")
    for c in e[4]:
        emit(c, "first")
    for c in cubic_add(F, e[4], e[3]):
        emit(c, "last")


def lookup_programs():
    """(constraints, map A, map B): Builders. Map A writes the three denominators (extended columns 0..8), map B the row sums
    (9..11) and a copy of them (12..14) for the prefix sum."""
    kt = LOOKUP_K0 + LOOKUP_K1
    c = Builder(CONSTRAINTS, kt, n_challenge=3)
    F = RecField(c)
    loc = [c.local(j) for j in range(kt)]
    nxt = [c.next(j) for j in range(kt)]
    beta = [c.challenge(j) for j in range(3)]
    lookup_constraints(F, loc, nxt, beta, lambda v, when: c.assert_zero(v, when))
    ma = Builder(MAP, kt, n_challenge=3, n_out_columns=LOOKUP_K1)
    beta = [ma.challenge(j) for j in range(3)]
    for j, col in enumerate((3, 4, 5)):
        ma.store(3 * j, ma.sub(beta[0], ma.local(col)))
        ma.store(3 * j + 1, beta[1])
        ma.store(3 * j + 2, beta[2])
    mb = Builder(MAP, kt, n_challenge=3, n_out_columns=LOOKUP_K1)
    G = RecField(mb)
    e = [tuple(mb.local(LOOKUP_K0 + 3 * j + k) for k in range(3)) for j in range(3)]
    row = cubic_sub(G, cubic_add(G, e[0], e[1]), cubic_scale(G, e[2], mb.local(6)))
    for k in range(3):
        mb.store(9 + k, row[k])
        mb.store(12 + k, row[k])
    return c, ma, mb


def lookup_steps(ma, mb):
    """the step list of cityprover.stark_desc / oracle_lib.stark_desc for the two map programs (as backend objects)"""
    return [("map", ma), ("cubic_inverse", 0, 3, CUBIC_MODULUS), ("map", mb), ("prefix_sum", 12, 3, True)]
