# LDS bank conflicts and LDS / VALU activity of the K2 super-panel kernels (one --pmc pass, no tracing flags)
root=$(pwd); out=gpurun_out/k2_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $root/$out/run -o p -- python $root/bench.py --only wepscale > $root/$out/run.log 2>&1
cd $root && python - <<'PY'
import csv, glob, re, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/k2_pmc/run/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
        if 'resid' in n: acc[n+" grid="+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k[:70], {c:int(sum(x)/len(x)) for c,x in v.items()})
PY
