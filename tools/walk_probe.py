"""Where a wave of a small launch spends its cycles: bh_force_walk_stats of the one-wave-per-group walk (small-launch
instance: prefetch, every stack entry in the lanes) — cycles per block, of which waiting for the block's records.
   python tools/walk_probe.py [n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhpkg  # noqa: E402

pkg = bhpkg.load()
for n in [int(a) for a in sys.argv[1:]] or [16384, 65536, 125000, 200000]:
    for g in (64, 32):
        e = pkg.Engine(n, force_coop=1, force_group=g)
        e.upload(*pkg.plummer(n, seed=42))
        e.step(3)
        e.tree_stages()
        ws = e.force_walk_stats()
        e.close()
        cyc = ws.wave_cycles_mean * ws.waves
        print(f"n={n} group={g}: waves {ws.waves} blocks/wave {ws.blocks / ws.waves:.0f} pairs/block {ws.pairs / ws.blocks:.2f} "
              f"clock {ws.clock_ghz:.2f} GHz | cycles/wave mean {ws.wave_cycles_mean:.0f} max {ws.wave_cycles_max:.0f} | "
              f"cycles/block {cyc / ws.blocks:.0f}, of which fetch wait {ws.fetch_wait_cycles / ws.blocks:.0f} "
              f"({ws.fetch_wait_cycles / cyc:.2f}); masked pairs {ws.masked_pairs / ws.pairs:.2f}, "
              f"nobody takes {ws.no_taker_pairs / ws.pairs:.3f}", flush=True)
