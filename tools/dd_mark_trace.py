"""design study: where a block of dd_mark_kernel spends its time (library built with -DBH_DD_TRACE: tools/mkvariant.sh
ddtrace -DBH_DD_TRACE; BH_LIB_PATH=tools/bin/libs/ddtrace.so).  Runs the in-process rehearsal (tools/dd_debug.py) and
prints the stamps of the LAST dd_mark launch.   python tools/dd_mark_trace.py [world] [n_total] [steps]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dd_debug  # noqa: E402
    dd_debug.main()
    lib = ctypes.CDLL(os.environ["BH_LIB_PATH"])
    buf = np.zeros((2048, 8), dtype=np.uint64)
    assert lib.bh_debug_dd_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.astype(np.int64)
    used = t[:, 0] > 0
    t = t[used]
    t0 = t[:, 0].min()
    print(f"blocks {len(t)}; first start .. last start {(t[:, 0].max() - t0) / 100:.1f} us; chunk loops end at "
          f"median {np.median(t[:, 5] - t0) / 100:.1f} max {(t[:, 5].max() - t0) / 100:.1f} us; last block's scan ends "
          f"{(t[:, 6].max() - t0) / 100:.1f} us")
    for k, lab in ((1, "collect candidates"), (2, "test candidates"), (3, "compact + publish")):
        print(f"   {lab:20s} per block: median {np.median(t[:, k]) / 100:6.1f}  p90 {np.percentile(t[:, k], 90) / 100:6.1f}  max {t[:, k].max() / 100:6.1f} us")
    print(f"   candidates per block: median {int(np.median(t[:, 4]))} max {t[:, 4].max()}")
    sys.exit(0)
world = sys.argv[1] if len(sys.argv) > 1 else "8"
n = sys.argv[2] if len(sys.argv) > 2 else "8000000"
steps = sys.argv[3] if len(sys.argv) > 3 else "4"
sys.exit(subprocess.call([sys.executable, __file__, "--child", "--world", world, "--n", n, "--steps", steps, "--no-split",
                          "--quiet"]))
