"""Same-box A/B of library builds over several body counts: whole-step time (best of 3 x 40 steps) and a hash of the
accelerations.   python tools/step_ab.py base <variant> ...   (variants: tools/mkvariant.sh -> tools/bin/libs/<name>.so)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, hashlib
import numpy as np
sys.path.insert(0, %r)
import bhpkg
pkg = bhpkg.load()
for n in (16384, 65536, 125000, 500000, 1000000):
    ic = pkg.plummer(n, seed=42)
    e = pkg.Engine(n)
    e.upload(*ic)
    e.step(10)
    e.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        e.step(40)
        e.sync()
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    acc = np.stack(e.download_acc(), 1)
    print(n, round(best, 4), hashlib.sha256(acc.tobytes()).hexdigest()[:8], flush=True)
    e.close()
""" % ROOT

for rep in range(2):
    for v in sys.argv[1:]:
        env = dict(os.environ)
        if v == "base":
            env.pop("BH_LIB_PATH", None)
        else:
            env["BH_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", "libs", v + ".so")
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        print(f"{v:10s}", " | ".join(out.stdout.strip().splitlines()), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])
