"""design study: time bbox..build alone (safe when the build is deliberately broken for an experiment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
e = pkg.Engine(n)
ic = pkg.plummer(n, seed=42)
for it in range(8):
    e.upload(*ic)
    e.bbox(); e.morton(); e.sort(); e.build()
e.sync()
s = e.stats(); print("cells", s.n_internal, "records", s.n_entries, "flags", s.status_flags)
