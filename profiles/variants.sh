#!/bin/bash
# Runs the headline workloads with each prebuilt library variant (build/variants/<name>/libportrayer_hip.so, made here
# with `make variant NAME=.. EXTRA_HIPFLAGS=..`). usage: bash profiles/variants.sh "name1 name2 .." ["workload args" ...]
VARS=$1; shift
cp portrayer_amd/libportrayer_hip.so /tmp/libportrayer_hip.so.keep
for v in current $VARS; do
  if [ $v = current ]; then cp /tmp/libportrayer_hip.so.keep portrayer_amd/libportrayer_hip.so; else cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; fi
  for wl in "$@"; do
    python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-10s %-32s %9.1f Mray/s %9.2f ms/frame' % ('$v', '$wl', d['value'], d['ms_per_step']))"
  done
done
cp /tmp/libportrayer_hip.so.keep portrayer_amd/libportrayer_hip.so
