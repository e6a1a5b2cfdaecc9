#!/bin/bash
# usage: scripts/prof_trace.sh <outdir-under-gpurun_out> <command...>   -> gpurun_out/<outdir>/kernel_trace.csv
set -e
out=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out/$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$root/gpurun_out/$out/raw" -o prof -- "$@" > "$root/gpurun_out/$out/cmd.log" 2>&1 || true
f=$(find "$root/gpurun_out/$out/raw" -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$root/gpurun_out/$out/kernel_trace.csv"; fi
rm -rf "$root/gpurun_out/$out/raw"
