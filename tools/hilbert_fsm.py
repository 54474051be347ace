"""Derives the finite-state form of the Hilbert numbering of csrc/bh_keys.h (Skilling's AxesToTranspose + bit
interleave): state x (2 bits of x, y, z) -> 6 key bits + next state, by probing the bit algorithm, checks it against
the bit algorithm on random and edge coordinates at full depth, and prints the table csrc/bh_hilbert_tab.h holds.
    python tools/hilbert_fsm.py [--emit]"""
import sys

import numpy as np

B = 21


def axes_to_transpose(x0, x1, x2):
    x0, x1, x2 = x0.copy(), x1.copy(), x2.copy()
    q = 1 << (B - 1)
    while q > 1:
        p = q - 1
        m = (x0 & q) != 0
        x0 = np.where(m, x0 ^ p, x0)
        for which in (1, 2):
            xi = x1 if which == 1 else x2
            m = (xi & q) != 0
            t = (x0 ^ xi) & p
            x0 = np.where(m, x0 ^ p, x0 ^ t)
            xi = np.where(m, xi, xi ^ t)
            if which == 1:
                x1 = xi
            else:
                x2 = xi
        q >>= 1
    x1 = x1 ^ x0
    x2 = x2 ^ x1
    t = np.zeros_like(x0)
    q = 1 << (B - 1)
    while q > 1:
        t = np.where((x2 & q) != 0, t ^ (q - 1), t)
        q >>= 1
    return x0 ^ t, x1 ^ t, x2 ^ t


def interleave(x0, x1, x2):
    k = np.zeros(x0.shape, dtype=np.uint64)
    for b in range(B):
        k |= ((x0 >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 2)
        k |= ((x1 >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 1)
        k |= ((x2 >> b) & 1).astype(np.uint64) << np.uint64(3 * b)
    return k


def hilbert_key(x, y, z):
    return interleave(*axes_to_transpose(x.astype(np.int64), y.astype(np.int64), z.astype(np.int64)))


def digits_below(prefix_levels, px, py, pz, depth):
    """key digits of levels [L, L + depth) for every combination of the next `depth` coordinate bits below the
    prefix (px, py, pz: L bits each)"""
    L = prefix_levels
    n = 8 ** depth
    idx = np.arange(n)
    # combination c: depth octants, most significant first; octant = (xb << 2) | (yb << 1) | zb
    xs = np.zeros(n, dtype=np.int64); ys = np.zeros(n, dtype=np.int64); zs = np.zeros(n, dtype=np.int64)
    for d in range(depth):
        o = (idx >> (3 * (depth - 1 - d))) & 7
        xs = (xs << 1) | ((o >> 2) & 1); ys = (ys << 1) | ((o >> 1) & 1); zs = (zs << 1) | (o & 1)
    sh = B - L - depth
    X = ((px << depth | xs) << sh); Y = ((py << depth | ys) << sh); Z = ((pz << depth | zs) << sh)
    k = hilbert_key(X, Y, Z)
    return ((k >> np.uint64(3 * sh)) & np.uint64(8 ** depth - 1)).astype(np.int64)


# ---- discover the states: signature = the digits of the next 2 levels for all 64 continuations
sig_to_state = {}
trans = []      # per state: list of 64 (6 key bits, next state) for two levels at once
pending = [(0, 0, 0, 0)]
state_prefix = []
while pending:
    L, px, py, pz = pending.pop(0)
    sig = tuple(digits_below(L, px, py, pz, 2))
    if sig in sig_to_state:
        continue
    sig_to_state[sig] = len(state_prefix)
    state_prefix.append((L, px, py, pz))
    if L + 4 <= B - 2:  # children two levels down
        for o in range(64):
            x2, y2, z2 = ((o >> 5) & 1) << 1 | ((o >> 2) & 1), ((o >> 4) & 1) << 1 | ((o >> 1) & 1), ((o >> 3) & 1) << 1 | (o & 1)
            pending.append((L + 2, px << 2 | x2, py << 2 | y2, pz << 2 | z2))
nstates = len(state_prefix)
print("states (two levels per step):", nstates, file=sys.stderr)
# table index: state * 64 + (xx << 4 | yy << 2 | zz) with xx = the two x bits (high first), etc.
tab = np.zeros((nstates, 64), dtype=np.int64)
for s, (L, px, py, pz) in enumerate(state_prefix):
    d2 = digits_below(L, px, py, pz, 2)  # indexed by (o_hi << 3 | o_lo)
    for xx in range(4):
        for yy in range(4):
            for zz in range(4):
                o_hi = ((xx >> 1) & 1) << 2 | ((yy >> 1) & 1) << 1 | ((zz >> 1) & 1)
                o_lo = (xx & 1) << 2 | (yy & 1) << 1 | (zz & 1)
                sig = tuple(digits_below(L + 2, px << 2 | xx, py << 2 | yy, pz << 2 | zz, 2))
                tab[s, xx << 4 | yy << 2 | zz] = (sig_to_state[sig] << 6) | int(d2[o_hi << 3 | o_lo])


def fsm_key(x, y, z):
    st = np.zeros(x.shape, dtype=np.int64)
    key = np.zeros(x.shape, dtype=np.uint64)
    for step in range(10):  # levels 0..19, two at a time
        sh = B - 2 - 2 * step
        e = tab[st, ((x >> sh) & 3) << 4 | ((y >> sh) & 3) << 2 | ((z >> sh) & 3)]
        key = (key << np.uint64(6)) | (e & 63).astype(np.uint64)
        st = e >> 6
    # the last level alone: use the two-level entry with the low bit as the HIGH bit of a pair and drop the low digit
    e = tab[st, ((x & 1) << 1) << 4 | ((y & 1) << 1) << 2 | ((z & 1) << 1)]
    return (key << np.uint64(3)) | ((e >> 3) & 7).astype(np.uint64)


rng = np.random.default_rng(7)
N = 2_000_000
x = rng.integers(0, 1 << B, N); y = rng.integers(0, 1 << B, N); z = rng.integers(0, 1 << B, N)
edge = np.array([0, 1, 2, (1 << B) - 1, (1 << B) - 2, 1 << 20, (1 << 20) - 1, 0x155555, 0xAAAAA])
ex, ey, ez = [a.ravel() for a in np.meshgrid(edge, edge, edge)]
x = np.concatenate([x, ex]); y = np.concatenate([y, ey]); z = np.concatenate([z, ez])
ok = np.array_equal(hilbert_key(x, y, z), fsm_key(x, y, z))
print("table reproduces the bit algorithm on", len(x), "coordinates:", ok, file=sys.stderr)
if not ok:
    sys.exit(1)
if "--emit" in sys.argv:
    print("// generated by tools/hilbert_fsm.py --emit: do not edit.  kHilbertTab[state * 64 + (xx << 4 | yy << 2 | zz)] =")
    print("// (next state << 7) | the six key bits of two levels — the next state as the BYTE offset of its 64 two-byte entries;")
    print("// xx = two bits of the x cell coordinate, high bit first (yy, zz alike)")
    print("#pragma once")
    print(f"constexpr int kHilbertStates = {nstates};")
    print(f"static __device__ const unsigned short kHilbertTab[{nstates * 64}] = {{")
    flat = ((tab >> 6) << 7 | (tab & 63)).ravel()
    for i in range(0, len(flat), 16):
        print("    " + ", ".join(str(int(v)) for v in flat[i:i + 16]) + ",")
    print("};")
