out=gpurun_out/r5a; mkdir -p $out
root=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr5 -o tr -- python $root/scripts/iar_runs.py 6 > $root/$out/trace_run.log 2>&1)
f=$(find /tmp/tr5 -name "*kernel_trace.csv" | head -1)
cp $f $out/iar_kernel_trace.csv
python scripts/diag/trace_steps.py $f 5 50 90 --setup > $out/r5_iar_steps_and_setup.txt 2>&1
python scripts/trace_k6.py $f > $out/r5_iar_trace_k6.json 2>&1
