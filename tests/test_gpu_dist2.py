"""Two processes, ONE GPU: the sharded contour drivers with device solves under world_size = 2.

RCCL refuses two ranks on one device, so the exchange of this test is staged through host memory and gloo
(`na.HostStagedComm`, same interface as the RCCL communicator of the C ABI).  Everything else is the production path of a
multi-GPU run: each rank takes the nodes i = r (mod 2), factorises them (host pool or the batched device LU when the pattern's
plan exists), solves on the GPU, accumulates its moments, exchanges, and extracts the eigenpairs -- both ranks must return the
eigenpairs of the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def na():
    import nep_amd
    assert nep_amd.device_count() >= 1, "no GPU visible"
    return nep_amd


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank(rank, world, port, out, n, device_lu):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["NEP_HOSTLU_WORKERS"] = "2"
    if not device_lu:
        os.environ["NEP_LU_DEV"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import nep_amd as na
    from nep_amd.linsolvers import _DeviceRefactor
    nep = na.nep_gallery("gun_spmf", n); nep.dev
    if device_lu:                                   # plan of the pattern: one host factorisation, then wait for the builder thread
        na.DeviceLU(nep.compute_Mder(250.0 ** 2 + 3.0))
        _DeviceRefactor.wait()
    na.MatrixTrapezoidalSharded.comm = na.HostStagedComm()
    info = {}
    Vh = na.probe_block(n, 16)
    lam, V = na.contour_beyn(nep, na.MatrixTrapezoidalSharded, sigma=250.0 ** 2, radius=1e4, N=32, k=16, neigs=10 ** 6, tol=1e-6,
                             Vh=Vh, sanity_check=True, info=info)
    used = sum(p["uses"] for p in _DeviceRefactor.plans.values())
    np.savez(os.path.join(out, "r%d.npz" % rank), lam=lam, V=V, nodes=info["nodes"], world=info["world"], used=used)
    na.HostLUPool.shutdown()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_lu", [False, True])
def test_sharded_beyn_two_ranks_one_gpu(na, tmp_path, device_lu):
    n = 1310
    mp.spawn(_rank, args=(2, _free_port(), str(tmp_path), n, device_lu), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz"); r1 = np.load(tmp_path / "r1.npz")
    assert int(r0["world"]) == 2 and int(r0["nodes"]) == 16 and int(r1["nodes"]) == 16
    assert np.array_equal(r0["lam"], r1["lam"]) and np.array_equal(r0["V"], r1["V"])          # identical on both ranks
    if device_lu:
        assert int(r0["used"]) >= 16 and int(r1["used"]) >= 16                               # every node of a rank on the device
    nep = na.nep_gallery("gun_spmf", n)
    lam, V = na.contour_beyn(nep, na.MatrixTrapezoidal, sigma=250.0 ** 2, radius=1e4, N=32, k=16, neigs=10 ** 6, tol=1e-6,
                             Vh=na.probe_block(n, 16), sanity_check=True)
    assert len(lam) == len(r0["lam"]) >= 1
    a = np.sort_complex(lam); b = np.sort_complex(r0["lam"])
    assert np.abs(a - b).max() <= 1e-8 * np.abs(a).max()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_two_ranks_rehearsal(na, world):
    """bench.py under torch.distributed.run with two (four; eight = the shape an 8-GPU node runs: 8 nodes per rank) ranks sharing the GPU (NEP_BENCH_SHARE_GPU=1: gloo process group, host-staged
    contour exchange): the N > 1 code path of the benchmark -- replicas of the headline step, max-over-ranks timing, ONE JSON
    line from rank 0, the sharded contour_beyn extra with 64 / world nodes per rank and its parity block"""
    import json
    import subprocess
    env = dict(os.environ, NEP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-wep-roofline", "--no-c3", "--no-c5", "--no-beyn-parity"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["eigenpairs_per_step"] >= 40
    b = d["beyn_sharded"]
    assert "error" not in b and b["nodes_per_rank"] == 64 // world and b["eigenpairs"] >= 20
    # per-rank record of the sharded extra: every rank reports its nodes, wall time and exchange time
    pr = b["per_rank"]
    assert [p["rank"] for p in pr] == list(range(world)) and all(p["nodes"] == 64 // world for p in pr)
    assert all(p["exchange_s"] is not None and p["exchange_s"] > 0 for p in pr)
    assert all(p.get("factorise_nodes_s") is not None and p["factorise_nodes_s"] > 0 for p in pr)
    assert abs(max(p["wall_s"] for p in pr) - b["seconds"]) < 0.05


def test_bench_gpus_flag_launches_its_own_ranks(na):
    """`python bench.py --gpus 2` as the driver invokes it -- NO torch.distributed.run around it: the file starts its two ranks itself
    (bench.launch_ranks) and rank 0 prints ONE line with n_gpus = 2"""
    import json
    import subprocess
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NEP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-wep-roofline", "--no-c3", "--no-c5", "--no-beyn-parity"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["beyn_sharded"]["nodes_per_rank"] == 32


def test_bench_headline_survives_a_failing_extra(na):
    """the sharded contour_beyn extra raises on EVERY rank (NEP_BENCH_BEYN_FAIL): rank 0 still prints one valid headline line,
    the failure is recorded under beyn_sharded.error and the process group shuts down cleanly"""
    import json
    import subprocess
    env = dict(os.environ, NEP_BENCH_SHARE_GPU="1", NEP_BENCH_BEYN_FAIL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-wep-roofline", "--no-c3", "--no-c5", "--no-beyn-parity"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["frac"] > 0
    assert "injected failure" in d["beyn_sharded"]["error"]


def test_sum_ranks_fixed_order_kernel(na):
    """the reduction half of nep_allgather_sum (nep_sum_ranks, csrc/comm.hip k_sum_ranks) on a FABRICATED world x len gather
    buffer: the result is the rank-ordered sum ((p0 + p1) + p2) + ... bit for bit -- what makes A0, A1 identical on every rank
    whatever the arrival order -- for world 1 .. 8, a length that is no multiple of the block size, aliasing output = block 0,
    and values whose sum depends on the order (so a tree or a reversed loop would be caught)"""
    from nep_amd._lib import lib, check, c_vp
    rng = np.random.default_rng(7)
    ln = 100003
    for world in (1, 2, 3, 8):
        P = (rng.standard_normal((world, ln)) * 10.0 ** rng.integers(-8, 8, (world, ln))
             + 1j * rng.standard_normal((world, ln)) * 10.0 ** rng.integers(-8, 8, (world, ln)))
        ref = P[0].copy()
        for r in range(1, world):
            ref = ref + P[r]
        G = torch.from_numpy(P).to("cuda").contiguous()
        out = torch.zeros(ln, dtype=torch.complex128, device="cuda")
        check(lib.nep_sum_ranks(c_vp(G.data_ptr()), ln, world, c_vp(out.data_ptr()), None))
        assert np.array_equal(out.cpu().numpy(), ref)
        if world >= 3:                                   # the order matters for these values: a reversed sum differs somewhere
            rev = P[world - 1].copy()
            for r in range(world - 2, -1, -1):
                rev = rev + P[r]
            assert not np.array_equal(rev, ref)
        check(lib.nep_sum_ranks(c_vp(G.data_ptr()), ln, world, c_vp(G.data_ptr()), None))          # output aliases block 0
        assert np.array_equal(G[0].cpu().numpy(), ref)


def _seed_rank(rank, world, port, out, n):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import nep_amd as na
    import nep_amd_hostlu as hl
    from nep_amd.linsolvers import _DeviceRefactor
    calls = [0]
    orig = hl.factor

    def counting(*a, **kw):
        calls[0] += 1
        return orig(*a, **kw)
    hl.factor = counting
    nep = na.nep_gallery("gun_spmf_scaled", n); nep.dev
    lu0 = na.seed_plan_from_rank0(nep, 0.0)                   # collective: rank 0 factorises, everybody gets the factors + a plan
    ready = [p["state"] for p in _DeviceRefactor.plans.values()]
    b = np.ones(n, dtype=complex)
    x0 = na.to_host(lu0.solve(na.to_dev(b)))
    lam, Q, _ = na.iar(nep, maxit=30, neigs=np.inf, v=np.ones(n), tol=1e-10)      # factorises M(0) on the device: no SuperLU here
    used = sum(p["uses"] for p in _DeviceRefactor.plans.values())
    np.savez(os.path.join(out, "s%d.npz" % rank), host_calls=calls[0], ready=np.array([r == "ready" for r in ready]), used=used, lam=lam, x0=x0)
    dist.destroy_process_group()


def test_seed_factorisation_broadcast_from_rank0(na, tmp_path):
    """the first (host) factorisation of a pattern is done by rank 0 only and broadcast (seed_plan_from_rank0): rank 1 never
    calls SuperLU, both ranks end up with a ready device-LU plan, use it in their next iar call and return the same eigenvalues;
    the solve with the broadcast factors is the same on both ranks"""
    n = 1310
    mp.spawn(_seed_rank, args=(2, _free_port(), str(tmp_path), n), nprocs=2, join=True)
    r0 = np.load(tmp_path / "s0.npz"); r1 = np.load(tmp_path / "s1.npz")
    assert int(r0["host_calls"]) == 1 and int(r1["host_calls"]) == 0
    assert r0["ready"].all() and r1["ready"].all() and len(r0["ready"]) == 1
    assert int(r0["used"]) >= 1 and int(r1["used"]) >= 1
    assert np.array_equal(r0["x0"], r1["x0"])
    assert len(r0["lam"]) == len(r1["lam"]) >= 1
    assert np.abs(np.sort_complex(r0["lam"]) - np.sort_complex(r1["lam"])).max() <= 1e-10 * np.abs(r0["lam"]).max()
