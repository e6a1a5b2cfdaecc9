"""world_size-2 test of the sharded quadrature (Beyn's multi-GPU seam) on CPU with the gloo backend:
node ownership i = r (mod P), one all-gather of the partial moment blocks, fixed-order sum ->
every rank holds bit-identical integrals that agree with the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _CpuOps:
    @staticmethod
    def axpy(alpha, x, y, length=None):
        y.add_(x, alpha=complex(alpha))

    @staticmethod
    def scal(x, alpha, length=None):
        x.mul_(complex(alpha))


def _f_factory(n, k):
    rng = np.random.default_rng(3)
    base = torch.from_numpy(rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n)))

    def f(t):
        return base * complex(np.cos(3 * t), np.sin(t)), complex(-np.sin(t), np.cos(t))
    return f


def _moments(K):
    """gv of the quadrature: K = 0 -> Beyn's two moments (1, g); K > 0 -> the 2K moments of contour_block_SS"""
    g = lambda t: complex(np.cos(t), np.sin(t))
    if K == 0:
        return [lambda s: 1.0 + 0j, g]
    return [(lambda s, kk=kk: g(s) ** kk) for kk in range(2 * K)]


def _run(rank, world, port, N, out, K=0, a=0.0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nep_amd as na
    info = {}
    S = na.integrate_interval(na.MatrixTrapezoidalSharded, _f_factory(50, 3), _moments(K), a,
                              a + 2 * np.pi, N, info=info, ops=_CpuOps)
    np.save(os.path.join(out, "S%d.npy" % rank), S.numpy())
    np.save(os.path.join(out, "nodes%d.npy" % rank), np.array([info["nodes"], info["world"]]))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("N,K,a", [(64, 0, 0.0), (7, 0, 0.0), (32, 3, np.pi / 32)])
def test_sharded_quadrature_world2(tmp_path, N, K, a):
    """Beyn's two moments, and the 2K = 6 moment blocks of contour_block_SS on the half-step-shifted nodes of its JSIAM
    mode, through the same seam"""
    import nep_amd as na
    world = 2
    mp.spawn(_run, args=(world, _free_port(), N, str(tmp_path), K, a), nprocs=world, join=True)
    S0 = np.load(tmp_path / "S0.npy"); S1 = np.load(tmp_path / "S1.npy")
    assert np.array_equal(S0, S1)                                   # bit-identical on all ranks
    n0 = np.load(tmp_path / "nodes0.npy"); n1 = np.load(tmp_path / "nodes1.npy")
    assert n0[1] == 2 and n0[0] + n1[0] == N and n0[0] == (N + 1) // 2
    Sref = na.integrate_interval(na.MatrixTrapezoidal, _f_factory(50, 3), _moments(K), a, a + 2 * np.pi,
                                 N, ops=_CpuOps).numpy()
    assert Sref.shape[0] == (2 if K == 0 else 2 * K)
    assert np.linalg.norm(S0 - Sref) <= 1e-13 * np.linalg.norm(Sref)


def test_sharded_needs_enough_nodes():
    import nep_amd as na
    # single process: world=1 -> fine even with N=1
    S = na.integrate_interval(na.MatrixTrapezoidalSharded, _f_factory(5, 2), [lambda s: 1.0 + 0j], 0.0, 1.0, 1,
                              ops=_CpuOps)
    assert S.shape == (1, 2, 5)
