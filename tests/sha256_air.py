"""A SHA-256 AIR for the generic AIR machinery (include/cityprover.h cp_air_* / cp_stark_prove, oracle/stark_air.c) — test
infrastructure. The reference proves SHA-256 with starkyx's `ByteStark` (city_common_circuit/src/hash/accelerator/sha256/
smartgadget.rs:55-79, :518-524) and its own test asserts exactly one thing about that proof: the digests it exposes equal SHA-256 of
the inputs (:505-513). The AIR of that STARK lives in an absent crate and is NOT restated here: this is an AIR of this repository's
own making — bit-decomposed words, one row per round, degree 3, no lookups — written ONCE over an abstract field (air_programs.py:
RecField records it into a program exactly as the Rust recording parser would, IntField checks it on rows, ExtField evaluates it at
zeta), so that the machinery is held on a REAL hash AIR whose trace satisfies it: proved, verified, digest == hashlib.sha256.

Layout: one row = one round, 64 rows = one block; a trace of 64 B rows hashes a padded message of B - 1 blocks, the last block is a
dummy whose only job is to carry the final chaining value on the last row, where it is constrained to the public digest.
Columns of a row (round r of its block), all values < 2^32 unless said otherwise:
  X[j][0..31]  bits of the eight working variables a..h BEFORE round r                                   256
  win[-1..14]  message words W[r-1] .. W[r+14] of the block (packed)                                       16
  w0[0..31]    bits of win[0] = W[r]      (every word passes position 0: that is its range check)          32
  w13[0..31]   bits of win[13] = W[r+13]                                                                   32
  sig0, sig1   sigma0(W[r]), sigma1(W[r+13]) packed                                                         2
  cw[0..1]     carry of the schedule sum W[r+15] = sigma1(W[r+13]) + W[r+8] + sigma0(W[r]) + W[r-1]         2
  ca[0..2], ce[0..2], cx[0..5]  carries of the eight state updates                                         12
  sel[0..63]   one-hot round selector, cyclic                                                              64
  H[0..7]      chaining value of the block (constant inside a block)                                        8
Constraints (degree <= 3): booleanity of every bit; packings; window shift and schedule inside a block; sigma definitions;
  pack(next X[j]) + 2^32 carry_j = out_j + sel[63] H[j]   with out = (T1 + T2, a, b, c, d + T1, e, f, g)   (transition)
  next H[j] = H[j] + sel[63] (pack(next X[j]) - H[j])                                                       (transition)
  first row: sel = e_0, X = H = IV; last row: H = public digest."""
import hashlib
import struct

import numpy as np

import air_programs as A

P = A.P
K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
     0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
     0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
     0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
     0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
     0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
M32 = 0xFFFFFFFF

# column offsets
X0 = 0                 # X[j][i] at X0 + 32 j + i
WIN = 256              # win[p] at WIN + p + 1, p = -1 .. 14
W0B = WIN + 16         # bits of win[0]
W13B = W0B + 32        # bits of win[13]
SIG = W13B + 32        # sig0, sig1
CW = SIG + 2           # 2 carry bits of the schedule
CA = CW + 2            # 3 carry bits of a
CE = CA + 3            # 3 carry bits of e
CX = CE + 3            # carry bits of b, c, d, f, g, h
SEL = CX + 6           # 64 selectors
HC = SEL + 64          # 8 chaining words
N_COLUMNS = HC + 8     # 424
N_PUBLIC = 8


def rotr(x, k): return ((x >> k) | (x << (32 - k))) & M32
def bits(x): return [(x >> i) & 1 for i in range(32)]


def pad(message):
    ml = len(message) * 8
    m = message + b"\x80" + b"\x00" * ((55 - len(message)) % 64) + struct.pack(">Q", ml)
    assert len(m) % 64 == 0
    return m


def max_message_bytes(log_rows):
    return ((1 << log_rows) // 64 - 1) * 64 - 9


def trace(message, log_rows):
    """(N_COLUMNS x 2^log_rows uint64 trace, the 8 digest words). The padded message must fill EXACTLY 2^log_rows / 64 - 1 blocks
    (a shorter one is refused), so that the chaining value on the last row is the digest of `message`; see max_message_bytes."""
    n = 1 << log_rows
    nb = n // 64
    m = pad(message)
    assert n >= 128 and len(m) // 64 == nb - 1, "the padded message must fill exactly %d blocks (it has %d)" % (nb - 1, len(m) // 64)
    m += b"\x00" * 64   # the dummy block
    t = np.zeros((N_COLUMNS, n), dtype=np.uint64)
    H = list(IV)
    for blk in range(nb):
        W = list(struct.unpack(">16I", m[64 * blk:64 * blk + 64]))
        for r in range(16, 64 + 15):
            s0 = rotr(W[r - 15], 7) ^ rotr(W[r - 15], 18) ^ (W[r - 15] >> 3)
            s1 = rotr(W[r - 2], 17) ^ rotr(W[r - 2], 19) ^ (W[r - 2] >> 10)
            W.append((W[r - 16] + s0 + W[r - 7] + s1) & M32)
        st = list(H)
        for r in range(64):
            row = 64 * blk + r
            a, b, c, d, e, f, g, h = st
            for j in range(8):
                t[X0 + 32 * j:X0 + 32 * j + 32, row] = bits(st[j])
            # window W[r-1] .. W[r+14]; position -1 of round 0 is never read by a constraint: 0. Words past W[63] are computed by the
            # same recurrence (the schedule constraint holds on every row 1..62 of a block whether the word is used or not)
            for p in range(-1, 15):
                t[WIN + p + 1, row] = W[r + p] if r + p >= 0 else 0
            t[W0B:W0B + 32, row] = bits(W[r])
            t[W13B:W13B + 32, row] = bits(W[r + 13])
            sg0 = rotr(W[r], 7) ^ rotr(W[r], 18) ^ (W[r] >> 3)
            sg1 = rotr(W[r + 13], 17) ^ rotr(W[r + 13], 19) ^ (W[r + 13] >> 10)
            t[SIG, row], t[SIG + 1, row] = sg0, sg1
            if 1 <= r <= 62:
                tot = sg1 + W[r + 8] + sg0 + W[r - 1]
                assert tot & M32 == W[r + 15]
                cw = tot >> 32
                t[CW, row], t[CW + 1, row] = cw & 1, cw >> 1
            S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)
            ch = (e & f) ^ (~e & g & M32)
            T1 = h + S1 + ch + K[r] + W[r]
            S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)
            mj = (a & b) ^ (a & c) ^ (b & c)
            T2 = S0 + mj
            out = [T1 + T2, a, b, c, d + T1, e, f, g]
            if r == 63:
                out = [o + Hj for o, Hj in zip(out, H)]
            new = [o & M32 for o in out]
            car = [o >> 32 for o in out]
            for i in range(3):
                t[CA + i, row] = (car[0] >> i) & 1
                t[CE + i, row] = (car[4] >> i) & 1
            assert car[0] < 8 and car[4] < 8
            for i, j in enumerate((1, 2, 3, 5, 6, 7)):
                assert car[j] < 2
                t[CX + i, row] = car[j]
            t[SEL + r, row] = 1
            t[HC:HC + 8, row] = H
            st = new
        H = st   # after round 63 `st` IS the new chaining value (`out` included H)
        if blk == nb - 2:
            digest = list(H)
    return t, digest


def digest_bytes(words):
    return struct.pack(">8I", *words)


# ---- the constraints, over an abstract field ------------------------------------------------------------------------------------
def _sum(F, xs):
    acc = xs[0]
    for x in xs[1:]:
        acc = F.add(acc, x)
    return acc


def _pack(F, bs, pw):
    """sum of bs[i] 2^i; pw = constants 2^0 .. 2^32"""
    return _sum(F, [F.mul(b, pw[i]) if i else b for i, b in enumerate(bs)])


def _xor2(F, two, x, y):
    return F.sub(F.add(x, y), F.mul(two, F.mul(x, y)))


def _xor3(F, two, four, x, y, z):
    xy, yz, zx = F.mul(x, y), F.mul(y, z), F.mul(z, x)
    return F.add(F.sub(_sum(F, [x, y, z]), F.mul(two, _sum(F, [xy, yz, zx]))), F.mul(four, F.mul(xy, z)))


def constraints(F, loc, nxt, pub, emit):
    """every constraint of the AIR over the field F. loc / nxt: the N_COLUMNS values of a row and of the next; pub: the 8 digest
    words; emit(value, when) with when in all / transition / first / last"""
    one, two, four = F.one, F.const(2), F.const(4)
    pw = [F.const(1 << i) for i in range(33)]
    X = [[loc[X0 + 32 * j + i] for i in range(32)] for j in range(8)]
    Xn = [[nxt[X0 + 32 * j + i] for i in range(32)] for j in range(8)]
    win = {p: loc[WIN + p + 1] for p in range(-1, 15)}
    winn = {p: nxt[WIN + p + 1] for p in range(-1, 15)}
    w0 = [loc[W0B + i] for i in range(32)]
    w13 = [loc[W13B + i] for i in range(32)]
    sig0, sig1 = loc[SIG], loc[SIG + 1]
    sel = [loc[SEL + r] for r in range(64)]
    seln = [nxt[SEL + r] for r in range(64)]
    H = [loc[HC + j] for j in range(8)]
    Hn = [nxt[HC + j] for j in range(8)]
    # booleanity
    for c in list(range(X0, X0 + 256)) + list(range(W0B, W0B + 64)) + list(range(CW, CW + 14)):
        v = loc[c]
        emit(F.mul(v, F.sub(v, one)), "all")
    # the two decomposed window words
    emit(F.sub(_pack(F, w0, pw), win[0]), "all")
    emit(F.sub(_pack(F, w13, pw), win[13]), "all")
    # sigma0(W[r]) = rotr 7 ^ rotr 18 ^ shr 3, sigma1(W[r+13]) = rotr 17 ^ rotr 19 ^ shr 10
    def small_sigma(b, r1, r2, sh):
        out = []
        for i in range(32):
            x, y = b[(i + r1) % 32], b[(i + r2) % 32]
            out.append(_xor3(F, two, four, x, y, b[i + sh]) if i + sh < 32 else _xor2(F, two, x, y))
        return _pack(F, out, pw)
    emit(F.sub(small_sigma(w0, 7, 18, 3), sig0), "all")
    emit(F.sub(small_sigma(w13, 17, 19, 10), sig1), "all")
    # inside a block the window moves by one word; the new last word is the schedule's (not after round 0: W[15] is message; not
    # after round 63: a new block begins)
    inside = F.sub(one, sel[63])
    for p in range(-1, 14):
        emit(F.mul(inside, F.sub(winn[p], win[p + 1])), "transition")
    sched = F.sub(F.sub(one, sel[0]), sel[63])
    cw = F.add(loc[CW], F.mul(two, loc[CW + 1]))
    total = _sum(F, [sig1, win[8], sig0, win[-1]])
    emit(F.mul(sched, F.sub(F.add(winn[14], F.mul(pw[32], cw)), total)), "transition")
    # the round
    a, b, c, d, e, f, g, h = X

    def big_sigma(v, r1, r2, r3):
        return _pack(F, [_xor3(F, two, four, v[(i + r1) % 32], v[(i + r2) % 32], v[(i + r3) % 32]) for i in range(32)], pw)
    ch = _pack(F, [F.add(g[i], F.mul(e[i], F.sub(f[i], g[i]))) for i in range(32)], pw)
    mj = []
    for i in range(32):
        ab = F.mul(a[i], b[i])
        mj.append(F.sub(_sum(F, [ab, F.mul(a[i], c[i]), F.mul(b[i], c[i])]), F.mul(two, F.mul(ab, c[i]))))
    kr = _sum(F, [F.mul(sel[r], F.const(K[r])) for r in range(64)])
    T1 = _sum(F, [_pack(F, h, pw), big_sigma(e, 6, 11, 25), ch, kr, win[0]])
    T2 = F.add(big_sigma(a, 2, 13, 22), _pack(F, mj, pw))
    packs = [_pack(F, X[j], pw) for j in range(8)]
    packn = [_pack(F, Xn[j], pw) for j in range(8)]
    out = [F.add(T1, T2), packs[0], packs[1], packs[2], F.add(packs[3], T1), packs[4], packs[5], packs[6]]
    ca = _sum(F, [F.mul(loc[CA + i], pw[i]) if i else loc[CA] for i in range(3)])
    ce = _sum(F, [F.mul(loc[CE + i], pw[i]) if i else loc[CE] for i in range(3)])
    car = [ca, loc[CX], loc[CX + 1], loc[CX + 2], ce, loc[CX + 3], loc[CX + 4], loc[CX + 5]]
    for j in range(8):
        emit(F.sub(F.add(packn[j], F.mul(pw[32], car[j])), F.add(out[j], F.mul(sel[63], H[j]))), "transition")
        emit(F.sub(Hn[j], F.add(H[j], F.mul(sel[63], F.sub(packn[j], H[j])))), "transition")
    # the round selector walks in a cycle from round 0
    for r in range(64):
        emit(F.sub(seln[(r + 1) % 64], sel[r]), "transition")
        emit(F.sub(sel[r], one) if r == 0 else sel[r], "first")
    for j in range(8):
        emit(F.sub(packs[j], F.const(IV[j])), "first")
        emit(F.sub(H[j], F.const(IV[j])), "first")
        emit(F.sub(H[j], pub[j]), "last")


def program():
    """the recorded constraint program (an air_programs.Builder)"""
    c = A.Builder(A.CONSTRAINTS, N_COLUMNS, n_public=N_PUBLIC)
    F = A.RecField(c)
    loc = [c.local(j) for j in range(N_COLUMNS)]
    nxt = [c.next(j) for j in range(N_COLUMNS)]
    pub = [c.public(j) for j in range(N_PUBLIC)]
    constraints(F, loc, nxt, pub, lambda v, when: c.assert_zero(v, when))
    return c


def random_message(seed, log_rows):
    rng = np.random.default_rng(seed)
    return bytes(rng.integers(0, 256, max_message_bytes(log_rows), dtype=np.uint8))


def check(message, log_rows):
    t, dg = trace(message, log_rows)
    assert digest_bytes(dg) == hashlib.sha256(message).digest()
    return t, dg
