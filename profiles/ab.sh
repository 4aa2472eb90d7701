#!/bin/bash
# A/B of kernel build variants on the GPU box: for each quoted flag set, rebuild the HIP library and
# run the bench. usage: bash profiles/ab.sh "<bench args>" "-DPT_MIN_WAVES=1" "-DPT_MIN_WAVES=2" ...
BARGS=$1; shift
for flags in "$@"; do
  rm -f portrayer_amd/libportrayer_hip.so portrayer_amd/csrc/pt_api.o
  make -s -j2 portrayer_amd/libportrayer_hip.so EXTRA_HIPFLAGS="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  for i in 1 2; do
    python bench.py --no-cpu-baseline $BARGS 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %8.1f Mray/s  %7.2f ms/frame  frac %.3f' % ('$flags', d['value'], d['ms_per_step'], d['roofline']['frac']))"
  done
done
