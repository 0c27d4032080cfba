"""Seeded synthetic circuits for the benchmark configurations:
a squaring chain  w_i * w_i = w_(i+1) - k_i  with m constraints.  Wires: [1, w_m (public output), w_0 .. w_(m-1)];
m = 2^k - 2 gives domainSize = 2^k (fake_setup.nim:203-206: ceilingLog2(m + npub + 1)).  Witness values are
full-width pseudo-random by construction (w_0 = 3, k_i from SplitMix64)."""
from __future__ import annotations

from . import bn128 as F
from .fake_setup import R1CS

R = F.primeR
_M64 = 0xFFFFFFFFFFFFFFFF


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def fr(self) -> int:
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


def _fr_stream(seed: int, count: int):
    """== [SplitMix64(seed).fr() for _ in range(count)], vectorised (the state is seed + (i+1)*golden)."""
    import numpy as np
    idx = np.arange(1, 4 * count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    raw = z.astype("<u8").tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") % R for i in range(count)]


def squaringChain(m: int, seed: int = 4, w0: int = 3):
    """-> (R1CS, witness values as ints)."""
    ks = _fr_stream(seed, m)

    def wire(i):
        return i + 2 if i < m else 1
    cons = []
    w = [0] * (m + 1)
    w[0] = w0 % R
    for i in range(m):
        w[i + 1] = (w[i] * w[i] + ks[i]) % R
        cons.append(([(wire(i), 1)], [(wire(i), 1)], [(wire(i + 1), 1), (0, (-ks[i]) % R)]))
    witness = [1, w[m]] + w[:m]
    return R1CS(nWires=m + 2, nPubOut=1, nPubIn=0, nPrivIn=1, constraints=cons), witness


def mixedCircuit(m: int, seed: int = 4, zero_pct: int = 40, one_pct: int = 30, lin_pct: int = 0, w0: int = 3):
    """A circuit whose witness looks like a circom witness (SURVEY 8d config 2-ii / config 5's shape) instead of the
    squaring chain's full-width values.  Same wire layout as squaringChain -- [1, v_m (public), v_0 .. v_(m-1)],
    constraint i defines v_(i+1) -- with four kinds of constraint, drawn per row from SplitMix64(seed + 1000):
      zero_pct %  boolean wire holding 0:  b * b = b
      one_pct  %  boolean wire holding 1:  b * b = b
      lin_pct  %  linear step  (cur + k_i) * 1 = new  -- `cur` enters A only, so a wire that is only ever used in
                  linear steps is ABSENT FROM B and its pointsB1 / pointsB2 entries are the point at infinity (0,0),
                  as snarkjs keys have them for such wires (zkey.nim loads them as they are, curves.nim:95-98)
      the rest    squaring step  cur * cur = new - k_i  (full-width values)
    The last row is always a squaring step (it defines the public output).  -> (R1CS, witness values as ints)."""
    ks = _fr_stream(seed, m)
    kind = SplitMix64(seed + 1000)

    def wire(i):
        return i + 2 if i < m else 1
    cons = []
    v = [0] * (m + 1)
    v[0] = w0 % R
    cur = 0                                    # index of the latest full-width value
    for i in range(m):
        u = kind.next() % 100 if i + 1 < m else 100
        new = wire(i + 1)
        if u < zero_pct + one_pct:
            v[i + 1] = 0 if u < zero_pct else 1
            cons.append(([(new, 1)], [(new, 1)], [(new, 1)]))
        elif u < zero_pct + one_pct + lin_pct:
            v[i + 1] = (v[cur] + ks[i]) % R
            cons.append(([(wire(cur), 1), (0, ks[i])], [(0, 1)], [(new, 1)]))
            cur = i + 1
        else:
            v[i + 1] = (v[cur] * v[cur] + ks[i]) % R
            cons.append(([(wire(cur), 1)], [(wire(cur), 1)], [(new, 1), (0, (-ks[i]) % R)]))
            cur = i + 1
    witness = [1, v[m]] + v[:m]
    return R1CS(nWires=m + 2, nPubOut=1, nPubIn=0, nPrivIn=1, constraints=cons), witness


# ---- BASELINE config 5's workload SHAPE: a Poseidon-shaped Merkle-inclusion circuit ------------------------------
# A real circom / snarkjs artefact cannot be produced without circom, snarkjs and a ptau file.  What CAN be built is a
# circuit with the same R1CS shape: x^5 S-boxes of 3 constraints each, the MDS mixing and the round constants inlined
# into the A / B linear combinations (what `circom -O2` does to circomlib's Ark / Mix templates), partial rounds whose
# un-S-boxed state stays a growing linear combination, path-selector bits and multiplexers -- rows of 1..~30 terms,
# ncoeffs >> n, a third of the wires absent from B.  NOT circomlib-compatible: round constants and MDS matrix are seeded
# pseudo-random field elements, and the hashes form one path of `depth` levels (one root, one public output).
class SparseR1CS:
    """An R1CS held as numpy triplets (row, wire, index into `values`) per matrix instead of Python lists: at 2^20
    constraints the A and B matrices hold ~10^7 non-zeros.  `constraints` (the list form of files/r1cs.nim:62-80 that
    fake_setup.R1CS carries) is derived on demand for small circuits (oracle and CPU tests)."""

    def __init__(self, nWires, nPubOut, nPubIn, nPrivIn, nConstraints, values, A, B, C):
        self.nWires, self.nPubOut, self.nPubIn, self.nPrivIn = nWires, nPubOut, nPubIn, nPrivIn
        self.nConstraints = nConstraints
        self.values = values            # list of ints (mod r): the distinct coefficient values
        self.A, self.B, self.C = A, B, C    # each: (row uint32[], wire uint32[], value index uint32[])
        self.wireToLabel = None

    @property
    def constraints(self):
        out = [([], [], []) for _ in range(self.nConstraints)]
        for k, (rows, wires, vi) in enumerate((self.A, self.B, self.C)):
            for r, w, v in zip(rows.tolist(), wires.tolist(), vi.tolist()):
                out[r][k].append((w, self.values[v]))
        return out

    def rowLengths(self, which="A"):
        import numpy as np
        return np.bincount(getattr(self, which)[0], minlength=self.nConstraints)


def _poseidon_params(seed, t=3, full=8, partial=57):
    rng = SplitMix64(seed)
    mds = [[rng.fr() for _ in range(t)] for _ in range(t)]
    ark = [[rng.fr() for _ in range(t)] for _ in range(full + partial)]
    return mds, ark


def poseidonMerkle(log2n: int, seed: int = 4, depth: int = None, cap: int = 24, full: int = 8, partial: int = 57):
    """-> (SparseR1CS, witness as ints).  domainSize = 2^log2n (fake_setup.nim:203-206) with as many Merkle levels as
    fit (or `depth`); wires [1, root (public), leaf, level 0 wires, level 1 wires, ...].  Per level: sibling s and
    path bit b (private inputs), b (b - 1) = 0, d = b (s - cur), then a width-3 Poseidon-shaped permutation of
    [0, cur + d, s - d]: `full` full rounds (half before, half after) and `partial` partial rounds; an S-box input is a
    linear combination of earlier wires (+ the round constant on wire 0), x2 = in * in, x4 = x2 * x2, x5 = x4 * in.
    In partial rounds the two un-S-boxed state elements stay linear combinations and grow by one term per round; a
    combination of more than `cap` terms becomes a wire (y = lc * 1).  The level's output (state 0) is a wire."""
    import numpy as np
    t = 3
    mds, ark = _poseidon_params(seed + 77, t, full, partial)
    ONE, PREV = -1, -2            # template wire ids: the constant wire, the previous level's output; >= 0: level-local
    # ---- one level as a template: rows of (A, B, C) linear combinations over template wire ids ----
    rows = []                     # (dictA, dictB, dictC)
    nloc = 0

    def new():
        nonlocal nloc
        nloc += 1
        return nloc - 1
    s_w, b_w, d_w = new(), new(), new()
    rows.append(({b_w: 1}, {b_w: 1, ONE: R - 1}, {}))                 # b (b - 1) = 0
    rows.append(({b_w: 1}, {s_w: 1, PREV: R - 1}, {d_w: 1}))          # d = b (s - cur)
    state = [{}, {PREV: 1, d_w: 1}, {s_w: 1, d_w: R - 1}]             # [capacity 0, left, right]
    # program for the witness generator: list of ops replayed numerically per level
    prog = [("sib",), ("bit",), ("mux",)]

    def lc_add(dst, src, k):
        for w, v in src.items():
            nv = (dst.get(w, 0) + v * k) % R
            if nv:
                dst[w] = nv
            else:
                dst.pop(w, None)

    def materialize(i):
        y = new()
        rows.append((dict(state[i]), {ONE: 1}, {y: 1}))
        state[i] = {y: 1}
        prog.append(("wire", i))
    for rnd in range(full + partial):
        is_full = rnd < full // 2 or rnd >= full // 2 + partial
        for i in range(t):
            lc_add(state[i], {ONE: 1}, ark[rnd][i])
        prog.append(("ark", rnd))
        for i in (range(t) if is_full else (0,)):
            x2, x4, x5 = new(), new(), new()
            lc = dict(state[i])
            rows.append((lc, dict(lc), {x2: 1}))
            rows.append(({x2: 1}, {x2: 1}, {x4: 1}))
            rows.append(({x4: 1}, dict(lc), {x5: 1}))
            state[i] = {x5: 1}
            prog.append(("sbox", i))
        mixed = []
        for i in range(t):
            acc = {}
            for j in range(t):
                lc_add(acc, state[j], mds[i][j])
            mixed.append(acc)
        state = mixed
        prog.append(("mix",))
        for i in range(1, t):      # state 0 is S-boxed (consumed) every round; the others accumulate
            if len(state[i]) > cap:
                materialize(i)
    out_w = new()
    rows.append((dict(state[0]), {ONE: 1}, {out_w: 1}))
    prog.append(("out",))
    rows_per_level, wires_per_level = len(rows), nloc
    # ---- size ----
    n = 1 << log2n
    npub = 1
    room = n - npub - 1 - 1                      # constraints <= n - npub - 1; one more row exports the root
    if depth is None:
        depth = room // rows_per_level
    assert depth >= 1 and depth * rows_per_level <= room, "domain too small for one level"
    ncons = depth * rows_per_level + 1
    assert ceilLog2(ncons + npub + 1) == log2n, "depth too small for this domain"
    base = 3                                     # wires 0 (one), 1 (root), 2 (leaf)
    nwires = base + depth * wires_per_level
    # ---- replicate the template ----
    values, vindex = [], {}

    def vid(v):
        if v not in vindex:
            vindex[v] = len(values)
            values.append(v)
        return vindex[v]
    mats = []
    for k in range(3):
        tr, tw, tv = [], [], []
        for r_, row in enumerate(rows):
            for w, v in row[k].items():
                tr.append(r_), tw.append(w), tv.append(vid(v))
        tr, tw, tv = np.array(tr, dtype=np.int64), np.array(tw, dtype=np.int64), np.array(tv, dtype=np.uint32)
        lv = np.arange(depth, dtype=np.int64)[:, None]
        R_ = (tr[None, :] + lv * rows_per_level)
        W_ = np.where(tw[None, :] >= 0, tw[None, :] + base + lv * wires_per_level, 0)
        prev = np.where(lv > 0, base + (lv - 1) * wires_per_level + out_w, 2)      # level 0 hashes the leaf
        W_ = np.where(tw[None, :] == PREV, prev, W_)
        V_ = np.broadcast_to(tv[None, :], R_.shape)
        mats.append([R_.reshape(-1).astype(np.uint32), W_.reshape(-1).astype(np.uint32), V_.reshape(-1).copy()])
    # root = out(last level) * 1
    last_out = base + (depth - 1) * wires_per_level + out_w
    extra = [([last_out], [vid(1)]), ([0], [vid(1)]), ([1], [vid(1)])]
    for k in range(3):
        mats[k][0] = np.concatenate([mats[k][0], np.full(len(extra[k][0]), ncons - 1, dtype=np.uint32)])
        mats[k][1] = np.concatenate([mats[k][1], np.array(extra[k][0], dtype=np.uint32)])
        mats[k][2] = np.concatenate([mats[k][2], np.array(extra[k][1], dtype=np.uint32)])
    r1cs = SparseR1CS(nwires, 1, 0, 1 + 2 * depth, ncons, values, tuple(mats[0]), tuple(mats[1]), tuple(mats[2]))
    r1cs.rowsPerLevel, r1cs.wiresPerLevel, r1cs.depth = rows_per_level, wires_per_level, depth
    # ---- witness: the same program, numerically ----
    rng = SplitMix64(seed)
    wit = [0] * nwires
    wit[0] = 1
    cur = rng.fr()
    wit[2] = cur
    for lv_ in range(depth):
        off = base + lv_ * wires_per_level
        nxt = 0                                    # next level-local wire
        st = None
        for op in prog:
            if op[0] == "sib":
                s_val = rng.fr()
                wit[off + nxt] = s_val
                nxt += 1
            elif op[0] == "bit":
                b_val = rng.next() & 1
                wit[off + nxt] = b_val
                nxt += 1
            elif op[0] == "mux":
                d_val = b_val * (s_val - cur) % R
                wit[off + nxt] = d_val
                nxt += 1
                st = [0, (cur + d_val) % R, (s_val - d_val) % R]
            elif op[0] == "ark":
                a_ = ark[op[1]]
                st = [(st[i] + a_[i]) % R for i in range(t)]
            elif op[0] == "sbox":
                x = st[op[1]]
                x2 = x * x % R
                x4 = x2 * x2 % R
                x5 = x4 * x % R
                wit[off + nxt], wit[off + nxt + 1], wit[off + nxt + 2] = x2, x4, x5
                nxt += 3
                st[op[1]] = x5
            elif op[0] == "mix":
                st = [sum(mds[i][j] * st[j] for j in range(t)) % R for i in range(t)]
            elif op[0] == "wire":
                wit[off + nxt] = st[op[1]]
                nxt += 1
            else:                                  # "out"
                wit[off + nxt] = st[0]
                cur = st[0]
                nxt += 1
        assert nxt == wires_per_level
    wit[1] = cur
    return r1cs, wit


def ceilLog2(x: int) -> int:
    return F.ceilingLog2(x)


def checkWitness(r1cs, wit, rows=None) -> bool:
    """(A z)(B z) == C z on every row (or on `rows`); plain Python ints -- for tests, not for 2^20"""
    cons = r1cs.constraints
    for i in (range(len(cons)) if rows is None else rows):
        a, b, c = (sum(v * wit[w] for w, v in lc) % R for lc in cons[i])
        if a * b % R != c:
            return False
    return True
