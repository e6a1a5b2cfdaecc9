#!/bin/bash
# usage: scripts/prof_stats.sh <outdir-under-gpurun_out> <command...>
# runs the command under rocprofv3 --kernel-trace --stats and leaves the kernel stats csv in gpurun_out/<outdir>/
set -e
out=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out/$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/$out/raw" -o prof -- "$@" > "$root/gpurun_out/$out/cmd.log" 2>&1 || true
f=$(find "$root/gpurun_out/$out/raw" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$root/gpurun_out/$out/kernel_stats.csv"; fi
rm -rf "$root/gpurun_out/$out/raw"
