"""GPU tests of the "reference mode" rows: the literal root-monopole force of the CUDA
binary, the viewer's colour mapping without GL, snapshot restart, and the C++ headless driver."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_literal_force_is_root_monopole(pkg):
    """literal_force=1: a = G*M*(COM-p)/(|COM-p|^2+eps2)^(3/2) with the reference's fp32 operation
    order (nbody_v5_bench.cu:205-213) — re-derived here with numpy float32 ops, bit for bit"""
    n = 5000
    ic = pkg.plummer(n, seed=13)
    e = pkg.Engine(n, literal_force=1)
    e.upload(*ic)
    e.tree_stages(); e.force()
    ax, ay, az = e.download_acc()
    root = e.download_tree()[0]
    f = np.float32
    dx = (f(root["x"]) - ic[0]).astype(f); dy = (f(root["y"]) - ic[1]).astype(f); dz = (f(root["z"]) - ic[2]).astype(f)
    d2 = ((dx * dx + dy * dy).astype(f) + dz * dz).astype(f)
    dist = np.sqrt((d2 + f(50.0)).astype(f)).astype(f)
    ff = ((f(0.5) * f(root["m"])) / ((dist * dist).astype(f) * dist).astype(f)).astype(f)
    assert np.array_equal(ax, (ff * dx).astype(f))
    assert np.array_equal(ay, (ff * dy).astype(f))
    assert np.array_equal(az, (ff * dz).astype(f))
    # and it really is different from Barnes-Hut
    e2 = pkg.Engine(n)
    e2.upload(*ic)
    e2.tree_stages(); e2.force()
    bx, _, _ = e2.download_acc()
    assert np.abs(bx - ax).max() > 1e-3 * np.abs(bx).max()
    e.close(); e2.close()


def test_export_visual_matches_viewer_formula(pkg):
    n = 3000
    ic = pkg.plummer(n, seed=14)
    e = pkg.Engine(n)
    e.upload(*ic)
    e.step(2)
    x, y, z, vx, vy, vz = e.download()
    pos, col = e.export_visual()
    assert np.array_equal(pos, np.stack([x, y, z], 1))
    f = np.float32
    speed = np.sqrt(((vx * vx + vy * vy).astype(f) + vz * vz).astype(f)).astype(f)
    t = np.minimum((speed / f(150.0)).astype(f), f(1.0))
    assert np.array_equal(col[:, 0], (f(0.4) + t * f(0.6)).astype(f))   # nbody_v5.cu:288-290
    assert np.array_equal(col[:, 1], (f(0.3) + t * f(0.4)).astype(f))
    assert np.array_equal(col[:, 2], (f(1.0) - t * f(0.7)).astype(f))
    e.close()


def test_snapshot_restart_continues_bit_exactly(pkg, tmp_path):
    """checkpoint after 3 steps, restore into a new context, 3 more steps == 6 steps straight through"""
    n = 20000
    ic = pkg.plummer(n, seed=15)
    a = pkg.Engine(n, theta=0.4)
    a.upload(*ic)
    a.step(3)
    snap = str(tmp_path / "s.bin")
    a.save_snapshot(snap)
    a.step(3)
    ref = np.stack(a.download(), 1)
    b = pkg.Engine.restore(snap)
    assert b.n == n and b.params.theta == np.float32(0.4)
    b.step(3)
    got = np.stack(b.download(), 1)
    assert np.array_equal(ref, got)
    assert np.array_equal(a.download_mass(), ic[6])
    txt = str(tmp_path / "s.txt")
    b.dump_text(txt)
    steps, x, *_ = pkg.read_text(txt)
    assert len(x) == n and np.abs(x - got[:, 0]).max() <= 1e-6 + 4e-7 * np.abs(got[:, 0]).max()
    a.close(); b.close()


def test_cpp_driver_prints_reference_table(tmp_path):
    """bh_bench = the reference's main(): same banner and `Frame | Trajanje (ms) | FPS` rows"""
    exe = os.path.join(ROOT, "nbody-barnes-hut-cuda_amd", "bh_bench")
    dump = str(tmp_path / "final.txt")
    r = subprocess.run([exe, "--n", "20000", "--steps", "5", "--warmup", "1", "--dump", dump],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    out = r.stdout.decode().splitlines()
    assert out[0] == "Pokretanje Benchmarka za N = 20000..."
    assert any(l.startswith("Frame      | Trajanje (ms)   | FPS") for l in out)
    rows = [l for l in out if l[:1].isdigit() and "|" in l]
    assert len(rows) == 5 and rows[0].split("|")[0].strip() == "0"
    lines = open(dump).read().splitlines()
    assert lines[2].startswith("# Bodies: 20000") and len(lines) == 4 + 20000
    # literal mode runs too
    r = subprocess.run([exe, "--n", "20000", "--steps", "2", "--literal-force", "--quiet"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
