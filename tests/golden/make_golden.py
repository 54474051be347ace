"""Regenerates tests/golden/plummer4096_seed42.{npz,json}.

The reference (bgcarmin/NBody-Barnes-Hut-CUDA) holds no golden vectors and cannot be built in
this image (CUDA + Thrust), so these vectors come from THIS repo's IC generator (libbh.so, host
function bh_ic_plummer) and CPU oracle; they pin both against drift across rounds and machines
and give the GPU tests a fixed target that does not need the oracle at all.  Parity with the
reference itself stays "unpinned" (see oracle/bh_oracle.h).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bhpkg  # noqa: E402
import oracle as O  # noqa: E402
from helpers import oracle_pipeline  # noqa: E402


def main():
    pkg = bhpkg.load()
    n = 4096
    ic = pkg.plummer(n, seed=42)
    p = O.params()
    o = oracle_pipeline(O, ic, p)
    codes, _ = O.morton30(*ic[:3], o["bounds"])
    acc, V, Oc, P = O.force(o["rec"], o["xyzm"], p, O.ORDER_PREORDER)
    acc_b, *_ = O.force(o["rec"], o["xyzm"], p, O.ORDER_BATCHED)
    nx, nv = O.integrate(o["xyzm"], o["vel"], acc, p)
    # 10 whole steps (batched order = the engine's), caller order
    st = O.Oracle(n, p)
    st.upload(*ic)
    st.step(10, order=O.ORDER_BATCHED)
    s10 = np.stack(st.download(), 1)
    out = dict(x=ic[0], y=ic[1], z=ic[2], vx=ic[3], vy=ic[4], vz=ic[5], m=ic[6],
               bounds=o["bounds"], sorted_keys=o["sorted_keys"], perm=o["perm"], morton30=codes,
               acc_preorder=acc, acc_batched=acc_b, V=V, O=Oc, P=P, xyzm_after=nx, vel_after=nv,
               state_after_10=s10)
    for f in ("kind", "first", "count", "s", "x", "y", "z", "m"):
        out["rec_" + f] = o["rec"][f]
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, "plummer4096_seed42.npz"), **out)
    meta = dict(n=n, seed=42, a=400.0, G=0.5, theta=0.5, eps2=50.0, dt=0.02, leaf_cap=1, key_bits=63,
                n_internal=int(o["n_internal"]), n_entries=int(len(o["rec"])),
                max_level=int(o["max_level"]), V_mean=float(V.mean()), O_mean=float(Oc.mean()),
                P_mean=float(P.mean()))
    json.dump(meta, open(os.path.join(here, "plummer4096_seed42.json"), "w"), indent=1)
    print(meta)


if __name__ == "__main__":
    main()
