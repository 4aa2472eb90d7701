#!/bin/bash
# v_readlane / v_writelane (scalar-register spill traffic: VALU issue slots) and scratch accesses per kernel of a HIP object.
#   bash profiles/tools/spill_lanes.sh portrayer_amd/csrc/pt_render_m3.o [name filter]
bash "$(dirname "$0")/kernel_resources.sh" "$1" > /dev/null 2>&1
B=$(basename "$1" .o)
awk -v filt="${2:-}" '/^[0-9a-f]+ <_Z/{name=$2} /v_readlane_b32/{r[name]++} /v_writelane_b32/{w[name]++} /scratch_/{s[name]++} /^\t/{n[name]++} END{for(k in n) if (filt=="" || index(k,filt)) printf "%-70s instrs %6d  v_readlane %4d  v_writelane %4d  scratch ops %4d\n", substr(k,1,70), n[k], r[k], w[k], s[k]}' /tmp/isa/$B.s | sort
