"""State dump / restart: host-only round trips of the text format of the reference's
older generation (/root/reference/output_bh.txt:1-4 — header reproduced from the survey, the file
itself is not read here) and of the binary snapshot."""
import os

import numpy as np
import pytest


def test_text_roundtrip_and_header(pkg, tmp_path):
    n = 1000
    x, y, z, vx, vy, vz, m = pkg.plummer(n, seed=3)
    path = str(tmp_path / "out.txt")
    assert pkg.write_text(path, 100, 0.5, 0.03, x, y, z, vx, vy, vz) == 0
    lines = open(path).read().splitlines()
    assert lines[0] == "# Barnes-Hut N-Body Simulation Results"
    assert lines[1] == "# Final positions and velocities after 100 steps"
    assert lines[2] == "# Bodies: 1000, Theta: 0.50, dt: 0.030"
    assert lines[3] == "# Format: x y z vx vy vz"
    assert len(lines) == 4 + n and len(lines[4].split()) == 6
    steps, rx, ry, rz, rvx, rvy, rvz = pkg.read_text(path)
    assert steps == 100
    # "%f" keeps 6 decimals, as the reference's dump does
    for a, b in ((x, rx), (y, ry), (z, rz), (vx, rvx), (vy, rvy), (vz, rvz)):
        assert np.abs(a - b).max() <= 1e-6 + 4e-7 * np.abs(a).max()


def test_text_reader_accepts_reference_style_rows(pkg, tmp_path):
    path = str(tmp_path / "ref_style.txt")
    with open(path, "w") as f:
        f.write("# Barnes-Hut N-Body Simulation Results\n# Final positions and velocities after 100 steps\n"
                "# Bodies: 2, Theta: 0.50, dt: 0.030\n# Format: x y z vx vy vz\n"
                "-231.422424 219.206589 45.071430 -2.259464 -2.246445 0.000000\n"
                "110.894081 271.127258 27.969097 -2.675597 1.189210 0.000000\n")
    steps, x, y, z, vx, vy, vz = pkg.read_text(path)
    assert steps == 100 and len(x) == 2
    assert x[0] == np.float32(-231.422424) and vz[1] == 0.0 and vy[1] == np.float32(1.189210)


def test_text_reader_rejects_truncated_file(pkg, tmp_path):
    path = str(tmp_path / "bad.txt")
    with open(path, "w") as f:
        f.write("# Bodies: 3, Theta: 0.50, dt: 0.030\n1 2 3 4 5 6\n")
    with pytest.raises(pkg.BhError):
        pkg.read_text(path)


def test_snapshot_roundtrip_is_lossless(pkg, tmp_path):
    import ctypes as C
    n = 777
    ic = pkg.plummer(n, seed=4)
    p = pkg.default_params(theta=0.3, leaf_cap=4)
    path = str(tmp_path / "snap.bin")
    F = C.POINTER(C.c_float)
    st = pkg.lib.bh_write_snapshot(path.encode(), n, 42, C.byref(p), *[a.ctypes.data_as(F) for a in ic])
    assert st == 0
    n2, steps, p2, arrs = pkg.read_snapshot(path)
    assert n2 == n and steps == 42 and p2.theta == np.float32(0.3) and p2.leaf_cap == 4
    for a, b in zip(ic, arrs):
        assert np.array_equal(a, b)
    # corrupt magic -> rejected
    raw = bytearray(open(path, "rb").read())
    raw[0] ^= 0xFF
    open(path, "wb").write(raw)
    with pytest.raises(pkg.BhError):
        pkg.read_snapshot(path)
    assert not os.path.exists(str(tmp_path / "missing.bin"))
    with pytest.raises(pkg.BhError):
        pkg.read_snapshot(str(tmp_path / "missing.bin"))
