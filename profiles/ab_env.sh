#!/bin/bash
# A/B of run-time settings (environment variables) with the current build.
# usage: bash profiles/ab_env.sh "<bench args>" "VAR=val ..." "VAR=val ..."
BARGS=$1; shift
for e in "$@"; do
  for i in 1 2; do
    env $e python bench.py --no-cpu-baseline $BARGS 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']['per_ray']
print('%-32s %8.1f Mray/s  %7.2f ms/frame  nodes/ray %.2f prim/ray %.2f' % ('$e', d['value'], d['ms_per_step'], r['inner_nodes'], r['primitive_tests']))"
  done
done
