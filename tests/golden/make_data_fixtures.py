"""Converts the reference's DATA files (matrices its own gallery/tests hold) into
compressed npz fixtures.  Run in the build container only (reads /root/reference):

    python tests/golden/make_data_fixtures.py

Writes tests/golden/gun_W.npz, tests/golden/qdep0.npz and a copy of gun_W.npz into the
product package's data directory (the product needs W1/W2 as problem input).
Sources: src/gallery_extra/converted_nlevp/gun_W{1,2}.txt,
         src/gallery_extra/converted_misc/qdep_infbilanczos_A{0,1}.txt
"""
import os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.gallery import read_sparse_matrix

REF = "/root/reference/src/gallery_extra"
HERE = os.path.dirname(os.path.abspath(__file__))


def pack(**mats):
    out = {}
    for k, A in mats.items():
        A = A.tocsc(); A.sum_duplicates(); A.sort_indices()
        out[k + "_data"] = A.data
        out[k + "_indices"] = A.indices.astype(np.int32)
        out[k + "_indptr"] = A.indptr.astype(np.int32)
        out[k + "_shape"] = np.array(A.shape, dtype=np.int64)
    return out


W1 = read_sparse_matrix(REF + "/converted_nlevp/gun_W1.txt")
W2 = read_sparse_matrix(REF + "/converted_nlevp/gun_W2.txt")
np.savez_compressed(HERE + "/gun_W.npz", **pack(W1=W1, W2=W2))
A0 = read_sparse_matrix(REF + "/converted_misc/qdep_infbilanczos_A0.txt")
A1 = read_sparse_matrix(REF + "/converted_misc/qdep_infbilanczos_A1.txt")
np.savez_compressed(HERE + "/qdep0.npz", **pack(A0=A0, A1=A1))
dst = os.path.join(HERE, "..", "..", "nonlineareigenproblems.jl_amd", "data")
os.makedirs(dst, exist_ok=True)
shutil.copy(HERE + "/gun_W.npz", dst + "/gun_W.npz")
shutil.copy(HERE + "/qdep0.npz", dst + "/qdep0.npz")
print("ok", W1.nnz, W2.nnz, A0.nnz, A1.nnz)
