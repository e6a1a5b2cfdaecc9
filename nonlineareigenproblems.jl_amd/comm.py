"""The multi-GPU exchange of the contour integrators through the C ABI (nep_comm_create / nep_allgather_sum, csrc/comm.hip):
one RCCL communicator per process (= per GPU), an all-gather of the partial moment block over xGMI and a sum in fixed rank
order.  The 128-byte unique id reaches the ranks out of band -- here through torch.distributed's process group (any backend),
in a Julia host through MPI.jl."""
import ctypes as C
import time

import numpy as np

from ._lib import lib, check, c_vp, hptr
from .nep import stream_ptr


class DeviceComm:
    def __init__(self, rank, world, unique_id):
        assert len(unique_id) == 128
        self.rank, self.world = int(rank), int(world)
        self._uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        h = c_vp()
        check(lib.nep_comm_create(self.rank, self.world, hptr(self._uid), C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id():
        uid = np.zeros(128, dtype=np.uint8)
        check(lib.nep_comm_unique_id(hptr(uid)))
        return uid.tobytes()

    _from_dist = None

    @classmethod
    def from_torch_distributed(cls):
        """communicator spanning the default torch.distributed group (created once per process)"""
        import torch.distributed as dist
        if cls._from_dist is None:
            rank, world = dist.get_rank(), dist.get_world_size()
            box = [cls.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            cls._from_dist = cls(rank, world, box[0])
        return cls._from_dist

    def allgather_sum(self, S, out=None):
        """out = sum over ranks of S (device complex128 tensor, same shape on every rank), identical on every rank"""
        out = S if out is None else out
        import torch
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.nep_allgather_sum(self.h, c_vp(S.data_ptr()), S.numel(), c_vp(out.data_ptr()), stream_ptr()))
        e1.record()
        self._last_events = (e0, e1)                  # read by last_exchange_s (no synchronisation here)
        return out

    @property
    def last_exchange_s(self):
        """device time of the last all-gather + sum (waits for it)"""
        ev = getattr(self, "_last_events", None)
        if ev is None:
            return None
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1]) * 1e-3

    def close(self):
        if getattr(self, "h", None):
            lib.nep_comm_destroy(self.h)
            self.h = None
            if DeviceComm._from_dist is self:
                DeviceComm._from_dist = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostStagedComm:
    """Same interface as DeviceComm, exchange staged through host memory and torch.distributed (any backend, e.g. gloo): for
    ranks that share ONE GPU -- RCCL refuses two ranks on a device -- i.e. for exercising the multi-rank sharded drivers with
    device solves on a single-GPU box (tests).  Fixed rank order of the sum, identical result on every rank."""

    def __init__(self):
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_sum(self, S, out=None):
        import torch
        import torch.distributed as dist
        out = S if out is None else out
        t0 = time.perf_counter()
        h = S.detach().cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h)
        # the same fixed-order sum kernel the RCCL path runs behind its all-gather (nep_sum_ranks): identical bits on every rank
        G = torch.stack(parts).to(S.device).contiguous()
        check(lib.nep_sum_ranks(c_vp(G.data_ptr()), S.numel(), self.world, c_vp(out.data_ptr()), stream_ptr()))
        torch.cuda.synchronize()
        self.last_exchange_s = time.perf_counter() - t0
        return out

    def close(self):
        pass
