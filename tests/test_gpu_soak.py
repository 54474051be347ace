"""Longer runs: the device-side status flags (record pool, traversal stack, sort look-back time-out,
domain-decomposition capacities) must stay clear over hundreds of steps and the state must stay finite."""
import numpy as np
import pytest

import bhpkg

pytestmark = pytest.mark.gpu


def test_single_context_300_steps_at_full_size():
    pkg = bhpkg.load()
    n = 1_000_000
    with pkg.Engine(n) as e:
        e.upload(*pkg.plummer(n, seed=17))
        for _ in range(3):
            e.step(100)
            st = e.stats()
            assert st.status_flags == 0 and st.n_entries <= 2 * n + 8
        x, y, z, vx, vy, vz = e.download()
    for a in (x, y, z, vx, vy, vz):
        assert np.isfinite(a).all()
    assert np.sqrt(vx * vx + vy * vy + vz * vz).max() <= 500.0 * (1 + 1e-6)   # ref:18 speed clamp


def test_domain_decomposed_150_steps():
    from test_gpu_dd import run_ranks, merge
    pkg = bhpkg.load()
    n = 400_000
    ic = pkg.plummer(n, seed=23)
    out = run_ranks(4, ic, 150)        # run_ranks asserts status_flags == 0 on every rank
    pos, vel, acc = merge(out, n)      # every body owned exactly once
    assert np.isfinite(pos).all() and np.isfinite(vel).all() and np.isfinite(acc).all()
    counts = np.array([o[-1] for o in out])
    assert np.abs(counts - n / 4).max() < 0.03 * n / 4, counts
    assert out[0][5] <= 1              # at most the initial LET retry
