"""CPU affinity: keep the host side of a rank on the NUMA node its GPU hangs off.

The host work of this backend (SuperLU factorisation, CSC->CSR conversions, pinned staging copies, k x k LAPACK) is
memory-bound and NUMA-sensitive: on a 2-socket EPYC 9575F box the same `DeviceLU(A)` takes 42 ms from a core of the GPU's
node, 70-100 ms from the other socket and anything in between when the scheduler migrates the process.  `pin_to_gpu_numa`
restricts every thread of the process to the GPU-local CPUs (read from sysfs); it never widens an existing affinity mask
and NEP_NO_PIN=1 disables it."""
import os

_done = {}


def _parse_cpulist(s):
    cpus = set()
    for part in s.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_local_cpus(device_index=0):
    """CPUs of the NUMA node of GPU `device_index` (empty set if sysfs does not tell)"""
    import torch
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            return _parse_cpulist(f.read())
    except Exception:
        return set()


def pin_to_gpu_numa(device_index=0):
    """returns the CPU set the process was restricted to (None if nothing was changed)"""
    if os.environ.get("NEP_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    if device_index in _done:
        return _done[device_index]
    res = None
    try:
        local = gpu_local_cpus(device_index)
        cur = os.sched_getaffinity(0)
        target = local & cur
        if target and target != cur:
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), target)
                except OSError:
                    pass
            res = target
    except Exception:
        res = None
    _done[device_index] = res
    return res


def cpu_budget():
    """CPUs this process may actually use: the smaller of its affinity mask and the cgroup CPU quota (cpu.max), divided
    by the number of ranks sharing the node (LOCAL_WORLD_SIZE / WORLD_SIZE).  Host-side thread and worker counts (eig
    workers of iar, SuperLU processes of contour_beyn) are sized from it: on the benchmark box the container is limited to
    16 CPUs' worth of time although 256 are visible, and oversubscribed workers only get throttled."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    ranks = 1
    for key in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        if os.environ.get(key, "").isdigit():
            ranks = max(1, int(os.environ[key]))
            break
    return max(1, n // ranks)
