out=gpurun_out/k2_ab; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_lds_dma scripts/ub/ub_lds_dma.hip 2>/dev/null && /tmp/ub_lds_dma > $out/ub_lds_dma.txt 2>&1; cat $out/ub_lds_dma.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "super_panel or resid_batch" > $out/pytest_k2.log 2>&1; tail -15 $out/pytest_k2.log
for m in 0 1; do NEP_K2_SP=$m timeout 600 python bench.py --only wepscale > $out/wepscale_sp$m.json 2> $out/wepscale_sp$m.err; done
python - <<'PY'
import json
for m in (0,1):
    try:
        d=json.loads([l for l in open('gpurun_out/k2_ab/wepscale_sp%d.json'%m) if l.startswith('{')][-1])
        print(m, {k:(round(v['ms_per_launch'],4), round(v['frac'],3)) for k,v in d.items() if isinstance(v,dict) and k.startswith('K2')})
    except Exception as e: print(m, 'ERR', e)
PY
