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
