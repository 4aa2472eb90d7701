#!/bin/bash
# VGPRs / spills / scratch of every render-kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: bash profiles/resources.sh [extra hipcc flags]
for m in 1 2 3 4 5 6 7; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-value "$@" -DPT_INST_MODE=$m \
      --cuda-device-only -Rpass-analysis=kernel-resource-usage -c portrayer_amd/csrc/pt_render_inst.hip -o /dev/null 2> /tmp/pt_res_$m.txt ) &
done
wait
for m in 1 2 3 4 5 6 7; do python3 - $m <<'PY'
import re, sys
t = open('/tmp/pt_res_%s.txt' % sys.argv[1]).read()
for blk in t.split('remark: Function Name: ')[1:]:
    name = blk.split()[0]
    k = re.search(r'pt_render_kernelILi(\d)ELb(\d)ELb(\d)ELi(\d)', name)
    if not k: continue
    g = lambda key: re.search(key + r': (\d+)', blk).group(1)
    print('mode %s stats %s tex %s park %s : vgpr %3s agpr %3s vgpr-spill %3s sgpr-spill %3s scratch %4s B/lane occupancy %s' % (*k.groups(), g(' VGPRs'), g('AGPRs'), g('VGPRs Spill'), g('SGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
PY
done
