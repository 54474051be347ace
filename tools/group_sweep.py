"""design study: force time for 32- vs 64-body groups (bh_params.force_group) across body counts"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
for n in [int(a) for a in sys.argv[1:]] or [65536, 98304, 131072, 196608, 262144, 524288]:
    ic = pkg.plummer(n, seed=42)
    row = []
    for g in (16, 32, 64):
        e = pkg.Engine(n, force_group=g)
        e.upload(*ic)
        e.set_timing(True)
        e.step(40)
        e.sync()
        f, t = e.timing_history()
        row.append((g, round(float(f[-30:].mean()), 4), round(float(t[-30:].mean()), 4)))
        e.close()
    print(n, row, flush=True)
