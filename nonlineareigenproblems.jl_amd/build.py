"""Builds libnepmi355.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The library is linked against the HIP runtime that PyTorch-ROCm itself loads
(torch/lib/libamdhip64.so), exactly as torch.utils.cpp_extension does, so that device pointers,
streams and RCCL communicators are shared with torch inside one process; /opt/rocm/lib is the
run-time fallback (rpath) for hosts without torch (e.g. the Julia binding).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnepmi355.so")
SOURCES = ["util.hip", "spmv.hip", "orth.hip", "gemm.hip", "trsv.hip", "trsv_ml.hip", "comm.hip", "driver.hip", "wep.hip", "lufac.hip"]


def _torch_lib_dir():
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + ["common.h", "trsv_ml.h"]] + [os.path.join(ROOT, "include", "nepmi355.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-Wno-unused-result"]
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + src)
        elif verbose and out:
            print(out.decode())
    # Link with the plain C++ driver (not hipcc) so that the HIP runtime dependency is the one we
    # name: torch's own libamdhip64.so when torch is present (one runtime per process), else ROCm's.
    tl = _torch_lib_dir()
    link = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", LIB] + objs
    if tl:
        link += ["-L" + tl, "-Wl,-rpath," + tl]
    link += ["-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
