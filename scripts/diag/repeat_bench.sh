#!/bin/bash
# usage: repeat_bench.sh <n> [ENV=VAL ...]   -- runs the short headline bench n times, reports failures
n=$1; shift
fail=0
for i in $(seq 1 $n); do
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-beyn --no-wep-roofline --no-c3 --no-c5 > /tmp/rb.out 2> /tmp/rb.err
  rc=$?
  nanw=$(grep -c "invalid value" /tmp/rb.err)
  if [ $rc -ne 0 ] || [ $nanw -ne 0 ]; then fail=$((fail+1)); echo "  run $i rc=$rc nan-warnings=$nanw $(grep -o "breakdown in step [0-9]*" /tmp/rb.err | head -1)"; fi
done
echo "config [$*]: $fail of $n runs bad"
