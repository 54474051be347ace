"""GPU tests of the multi-rank plumbing that can run on ONE MI355X: the RCCL code path of bench.py at
world size 1 (init, sharded stepper, all-gather into the bound acc buffer), and ShardedStepper ==
bh_step on the device.  Real multi-GPU runs belong to the driver; the N > 1 logic itself is covered
on CPU by tests/test_dist_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,word", [("domain", "domain decomposition"), ("replicated", "sharded")])
def test_bench_distributed_path_world1(mode, word):
    env = dict(os.environ)
    env["BH_FORCE_DIST"] = "1"
    env["BH_DIST_MODE"] = mode
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--bodies", "200000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["n_total"] == 200000
    assert word in out["config"]["parallelism"]
    # the line says what it measured: the metric names THIS workload, the 1-GPU reference of the scaling figures is
    # measured in the same run, and (domain scheme) the strong-scaling run BASELINE.json's metric is worded on rides along
    assert "200,000 bodies per GPU" in out["metric"] and "theta=0.5" in out["metric"]
    assert out["n1_ms_per_step"] > 0 and out["n1"]["bodies"] == 200000
    assert abs(out["aggregate_x"] - out["n1_ms_per_step"] / out["ms_per_step"]) < 1e-6 * out["aggregate_x"]
    assert len(out["build"]["csrc_sha16"]) == 16 and out["build"]["abi"] == 6
    if mode == "domain":
        assert "library RCCL transport" in out["config"]["parallelism"]      # ncclCommInitRank at world size 1
        assert out["strong"]["n_total"] == 200000 and out["strong"]["value"] > 0 and out["strong"]["scaling"] == "strong"
        assert out["config"]["domain"]["phase_ms_rank0"]["top_remote_force"] > 0


@pytest.mark.parametrize("mode", ["domain", "replicated"])
def test_bench_two_ranks_rehearsal(mode):
    """bench.py end to end with WORLD_SIZE=2 (both ranks on this one GPU, gloo instead of RCCL): the
    N > 1 code path of the benchmark itself — IC generation for 2 x bodies, stepper, timing reduction,
    JSON assembly.  The throughput of such a run is meaningless and not checked."""
    env = dict(os.environ)
    env["BH_BENCH_REHEARSAL"] = "1"
    env["BH_DIST_MODE"] = mode
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--bodies", "150000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["n_total"] == 300000 and out["value"] > 0
    assert "domain_fallback" not in out["config"]
    assert "150,000 bodies per GPU" in out["metric"] and out["aggregate_x"] > 0
    if mode == "domain":
        assert len(out["config"]["domain"]["let_records_per_rank"]) == 2
        assert out["strong"]["n_total"] == 150000 and len(out["strong"]["let_records_per_rank"]) == 2
        # the line says which force form ran and — adaptive form, ranks large enough for it — what X4 was measured at
        dom = out["config"]["domain"]
        assert dom["force_passes"].startswith("adaptive") and dom["split_now_rank0"] in (0, 1)
        assert "x4_ms_measured_rank0" in dom   # (null here: 150,000 bodies per rank never split)


def test_bench_two_ranks_domain_fallback_is_collective():
    """a LET that cannot fit (tiny let_cap) makes DomainStepper raise DomainLeft on EVERY rank after the same
    exchange; bench.py answers it by moving both ranks to the replicated scheme together, reports
    `domain_fallback`, and still finishes in lockstep (no rank is left inside a collective)"""
    env = dict(os.environ)
    env["BH_BENCH_REHEARSAL"] = "1"
    env["BH_DIST_MODE"] = "domain"
    env["BH_BENCH_LET_CAP"] = "2048"
    env["BH_COLLECTIVE_TIMEOUT_S"] = "120"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--bodies", "100000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "DomainLeft" in out["config"]["domain_fallback"]
    assert "replicated" in out["config"]["parallelism"]


@pytest.mark.parametrize("world,split,pct", [(2, "one", 0), (3, "one", 0), (2, "two", 0), (3, "two", 100), (3, "two", 50)])
def test_domain_stepper_multiprocess_one_gpu(world, split, pct, tmp_path):
    """the real multi-process flow of the domain-decomposed step (torch.distributed, one process per rank, every rank
    with its OWN main and side stream — the in-process tests serialise a rank's launches on one stream) with `world`
    ranks sharing this one GPU over the gloo backend; one pass, the partial two-pass form (own pass of the
    first 30 % behind the LET export, beside X4, then their remote pass beside the one pass of the rest: two launches at
    once that share one fold) and every body in two passes; rank 0 compares the gathered state with a single-context
    run (tests/dd_gpu_worker.py)"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = tmp_path / "dd.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dd_gpu_worker.py"), str(out), "60000", "6", split, str(pct)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(out.read_text())
    assert res["world"] == world and res["owned_once"] and res["flags"] == 0
    assert res["max_dpos"] < 5e-2, res
    assert res["acc_rel_median"] < 1e-4, res


def test_domain_stepper_multiprocess_large_ranks_two_launches_at_once(tmp_path):
    """2 processes x 500,000 bodies: the sizes at which the one pass of the unsplit 70 % is a MIXED launch that starts
    at a group offset (force_mixed_kernel g0) on the main stream while the remote pass of the first 30 % runs on the
    side stream — really at once here (own streams per process) — and both integrate into one shared fold"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = tmp_path / "dd.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dd_gpu_worker.py"), str(out), "1000000", "4", "two", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(out.read_text())
    assert res["world"] == 2 and res["owned_once"] and res["flags"] == 0
    assert res["max_dpos"] < 5e-2, res
    assert res["acc_rel_median"] < 1e-4, res


def test_adaptive_form_turns_the_split_on_when_the_exchange_is_slow(tmp_path):
    """bh_rank_opts.split 2 (what bench.py uses under torchrun): the rank times X4 with events on its stream and walks
    part of its bodies in two passes once the exchange lasts ~0.2 ms and more.  Here X4 goes through gloo (device ->
    host -> TCP -> device: milliseconds), so after the first few steps of 2 x 500,000 bodies the split form must be
    on, the measured duration reported, and the state still that of the single-context run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = tmp_path / "dd.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dd_gpu_worker.py"), str(out), "1000000", "8", "adaptive", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads(out.read_text())
    assert res["world"] == 2 and res["owned_once"] and res["flags"] == 0
    assert res["x4_us"] > 200 and res["split_now"] == 1, res
    assert res["max_dpos"] < 5e-2, res
    assert res["acc_rel_median"] < 1e-4, res


def test_bh_bench_whole_node_frame_loop_on_one_gpu():
    """the C++ host of the multi-GPU step (host/bh_bench.cpp --gpus / --devices: bh_create_group, bh_group_upload,
    bh_step_group, bh_group_sync; the frame loop of ref:353-367 for a whole node): 4 ranks on this one GPU through the
    in-process transport, and the RCCL path (ncclCommInitAll) at world size 1"""
    exe = os.path.join(ROOT, "nbody-barnes-hut-cuda_amd", "bh_bench")
    for extra in (["--devices", "0,0,0,0"], ["--gpus", "1", "--dist"]):
        r = subprocess.run([exe, "--n", "200000", "--steps", "4", "--warmup", "2", "--ic", "plummer"] + extra,
                           cwd=ROOT, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:] + r.stdout.decode()[-2000:]
        txt = r.stdout.decode()
        assert "Pokretanje Benchmarka za N = 200000" in txt and "Trajanje (ms)" in txt     # ref:287, 351
        assert f"gpus={4 if 'devices' in extra[0] else 1} " in txt and "particles/s/step" in txt
        assert "LET retries" in txt


def test_sharded_stepper_equals_bh_step(pkg):
    """stage calls + bh_force_range + bound acc buffer (the multi-rank step) == bh_step"""
    import torch
    from nbody_barnes_hut_cuda_amd import dist as bhdist
    n = 50000
    ic = pkg.plummer(n, seed=5)
    e1 = pkg.Engine(n)
    e1.upload(*ic)
    e1.step(4)
    ref = np.stack(e1.download(), 1)
    e1.close()
    e2, stepper = bhdist.make_gpu_stepper(pkg, n, device=0)
    e2.upload(*ic)
    stepper.step(4)
    torch.cuda.synchronize()
    got = np.stack(e2.download(), 1)
    assert np.array_equal(ref, got)
    e2.close()
