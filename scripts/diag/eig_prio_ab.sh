# A/B of the batch plan of the device eigen-decompositions on the headline call (scripts/iar_runs.py): size of the last batch, largest batch
out=gpurun_out/eig_ab; mkdir -p $out
for cfg in "NEP_IAR_EIG_LAST=8" "NEP_IAR_EIG_LAST=10" "NEP_IAR_EIG_LAST=12" "NEP_IAR_EIG_LAST=8 NEP_IAR_EIG_BATCH=12" "NEP_IAR_EIG_LAST=10 NEP_IAR_EIG_BATCH=12" "NEP_IAR_EIG_LAST=8 NEP_IAR_EIG_MS100=4.5" "NEP_IAR_EIG_LAST=6 NEP_IAR_EIG_MS100=4.5"; do
  env $cfg python scripts/iar_runs.py 14 2>/dev/null | tail -10 | awk -v c="$cfg" '{s+=$3; n++} END {printf "%-50s mean of last 10 calls %.2f ms\n", c, s/n}'
done
