import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nep_amd as na
which = sys.argv[1] if len(sys.argv)>1 else "wep"
if which=="wep":
    from nep_amd import wep
    wd=wep.WaveguideData(1003,999,"JARLEBRING"); dev=na.SPMFDevice(wd.big_matrices()); n=wd.n; ks=(1,60)
else:
    nep=na.nep_gallery("gun_spmf_scaled"); dev=nep.dev; n=nep.n; ks=(1,100)
for k in ks:
    V=torch.complex(torch.randn(k,n,dtype=torch.float64,device="cuda"),torch.randn(k,n,dtype=torch.float64,device="cuda"))
    Cdev=na.to_dev(np.random.default_rng(0).standard_normal((k,dev.mt))+0j)
    z=torch.empty(n,dtype=torch.complex128,device="cuda")
    for _ in range(10): dev.mlincomb_dev(Cdev,k,k,V,n,z)
    torch.cuda.synchronize()
    print(which,"k",k,"algorithmic bytes",dev.algorithmic_bytes(k))
