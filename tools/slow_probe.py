"""where do splitter-sort buckets overflow LDS? (bh_stats.sort_slow_buckets after each phase of a bench-like run)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhpkg
pkg = bhpkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ic = pkg.plummer(n, seed=42)
e = pkg.Engine(n)
e.upload(*ic)

def slow(tag):
    # raw counter WITHOUT the side effect on the sort choice: read the struct, then undo nothing (we only print)
    st = e.stats()
    print(f"{tag:34s} steps {st.steps:4d} slow buckets {st.sort_slow_buckets}", flush=True)
    return st.sort_slow_buckets

slow("after upload")
e.step(1); slow("after step 1 (radix)")
for k in range(6):
    e.step(1); slow(f"after step {k+2}")
e.tree_stages(); slow("after tree_stages")
e.force_count(); slow("after force_count")
e.step(5); slow("after 5 more steps")
e.close()
