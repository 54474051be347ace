"""design study: time the sort stage alone (bbox, morton, sort in a loop); safe with a broken sort"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
e = pkg.Engine(n, sort_variant=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ic = pkg.plummer(n, seed=42)
e.upload(*ic)
for it in range(30):
    e.upload(*ic) if it == 0 else None
    e.bbox(); e.morton(); e.sort()
e.sync()
print("flags", e.stats().status_flags)
