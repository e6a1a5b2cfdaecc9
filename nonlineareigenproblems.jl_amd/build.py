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
SOURCES = ["util.hip", "spmv.hip", "spmv_tile.hip", "orth.hip", "gemm.hip", "trsv.hip", "trsv_ml.hip", "comm.hip", "driver.hip", "wep.hip", "lufac.hip", "hesseig.hip", "iar_run.hip"]


def _torch_lib_dir():
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


DIGEST_FILE = LIB + ".digest"


def _deps():
    return [os.path.join(CSRC, f) for f in SOURCES + ["common.h", "trsv_ml.h", "devprims.h"]] + [os.path.join(ROOT, "include", "nepmi355.h")]


def source_digest():
    """sha256 (first 16 hex digits) over the CONTENTS of every source the library is built from: staleness must not depend on
    modification times (a fresh checkout or a copied tree has arbitrary ones)"""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_digest():
    try:
        with open(DIGEST_FILE) as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or built_digest() != source_digest()


class _BuildLock:
    """exclusive lock around check + compile + link: the ranks of a multi-process run that all find the library stale
    build it ONCE (the others wait, re-check under the lock and load the finished file)"""

    def __enter__(self):
        import fcntl
        os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
        self.f = open(os.path.join(HERE, "build", ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    with _BuildLock():
        if not force and not needs_build():          # another process built it while this one waited for the lock
            return LIB
        return _build_locked(verbose)


def _build_locked(verbose):
    digest = source_digest()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-Wno-unused-result", '-DNEP_SRC_DIGEST="%s"' % digest]
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
    tmp_lib = LIB + ".tmp.%d" % os.getpid()          # link beside the target, then rename: a reader never sees a half-written file
    link = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", tmp_lib] + objs
    if tl:
        link += ["-L" + tl, "-Wl,-rpath," + tl]
    link += ["-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(link), flush=True)
    try:
        subprocess.check_call(link)
        os.replace(tmp_lib, LIB)
    finally:
        if os.path.exists(tmp_lib):
            os.unlink(tmp_lib)
    with open(DIGEST_FILE + ".tmp", "w") as f:
        f.write(digest + "\n")
    os.replace(DIGEST_FILE + ".tmp", DIGEST_FILE)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
