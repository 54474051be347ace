"""Shared helpers for the parity tests: run the oracle pipeline stage by stage."""
import numpy as np


def oracle_pipeline(O, ic, p):
    """Oracle stages on caller-order arrays -> dict of every intermediate."""
    x, y, z, vx, vy, vz, m = ic
    b = O.bbox(x, y, z)
    k = O.keys(x, y, z, b, p.key_bits, p.key_curve)
    sk, perm = O.sort(k)
    xyzm = np.stack([x, y, z, m], 1)[perm].astype(np.float32)
    vel = np.stack([vx, vy, vz], 1)[perm].astype(np.float32)
    rec, lo, hi, ni, ml = O.build_tree(sk, p, O.root_edge(b))
    rec = O.com(rec, lo, hi, xyzm)
    return dict(bounds=b, keys=k, sorted_keys=sk, perm=perm, xyzm=xyzm, vel=vel, rec=rec, er_lo=lo,
                er_hi=hi, n_internal=ni, max_level=ml)


def key_curve_of(gp):
    """the curve an engine with these bh_params orders its keys by (30-bit keys are always Morton)"""
    return 0 if gp.key_bits == 30 else gp.key_curve


def oparams(O, gp):
    """oracle params from an engine's bh_params"""
    return O.params(G=gp.G, theta=gp.theta, dt=gp.dt, eps2=gp.eps2, max_speed=gp.max_speed,
                    leaf_cap=gp.leaf_cap, max_depth=gp.max_depth, key_bits=gp.key_bits,
                    key_curve=key_curve_of(gp))


def special_ics(name, n, rng):
    """Edge-case inputs the reference never tests (it has no tests): SURVEY §4."""
    f = np.float32
    vx = rng.normal(0, 1, n).astype(f)
    vy = rng.normal(0, 1, n).astype(f)
    vz = rng.normal(0, 1, n).astype(f)
    m = (2 + 5 * rng.random(n)).astype(f)
    if name == "coincident":
        x = np.full(n, 3.25, f); y = np.full(n, -7.5, f); z = np.full(n, 11.0, f)
    elif name == "collinear":
        x = np.linspace(-100, 100, n).astype(f); y = np.zeros(n, f); z = np.zeros(n, f)
    elif name == "outlier":
        x = rng.normal(0, 1, n).astype(f); y = rng.normal(0, 1, n).astype(f); z = rng.normal(0, 1, n).astype(f)
        x[0] = 1e6
    elif name == "pairs":  # near-coincident pairs -> deep chains
        h = n // 2
        bx = rng.uniform(-500, 500, h); by = rng.uniform(-500, 500, h); bz = rng.uniform(-500, 500, h)
        x = np.concatenate([bx, bx + 1e-3, np.zeros(n - 2 * h)]).astype(f)
        y = np.concatenate([by, by, np.zeros(n - 2 * h)]).astype(f)
        z = np.concatenate([bz, bz, np.zeros(n - 2 * h)]).astype(f)
    elif name == "tiny":  # cloud smaller than 1 unit: size clamp fmaxf(.,1) (ref:55)
        x = rng.uniform(0, 0.1, n).astype(f); y = rng.uniform(0, 0.1, n).astype(f); z = rng.uniform(0, 0.1, n).astype(f)
    elif name == "grid":  # bodies exactly on cell faces
        g = int(round(n ** (1 / 3))) or 1
        idx = np.arange(n)
        x = ((idx % g) * 16.0).astype(f); y = (((idx // g) % g) * 16.0).astype(f); z = ((idx // (g * g)) * 16.0).astype(f)
    elif name == "zero_mass":
        x = rng.uniform(-100, 100, n).astype(f); y = rng.uniform(-100, 100, n).astype(f); z = rng.uniform(-100, 100, n).astype(f)
        m[::3] = 0.0
    else:
        raise KeyError(name)
    return x, y, z, vx, vy, vz, m
