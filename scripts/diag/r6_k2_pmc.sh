# stall / occupancy counters of the K2 super-panel kernels at waveguide scale: one --pmc pass per counter group
# (no tracing flags), bench.py --only wepscale as the workload; summary -> profiles/pmc2/r6_k2_stall.json
root=$(pwd); out=gpurun_out/r6_k2_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $root/$out/g$i -o p -- python $root/bench.py --only wepscale > $root/$out/g$i.log 2>&1
done
cd $root && python - <<'PY'
import csv, glob, re, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r6_k2_pmc/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
        if 'tile_resid' in n: acc[n+" grid="+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:sum(x)/len(x) for c,x in v.items()} for k,v in acc.items()}
json.dump(out, open('gpurun_out/r6_k2_pmc/summary.json','w'), indent=1)
for k,v in out.items(): print(k[:90], {c:int(x) for c,x in v.items()})
PY
