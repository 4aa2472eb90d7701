#!/bin/bash
# Per-kernel register / scratch / LDS use of a HIP object (reads the code object's metadata; no GPU needed).
#   bash profiles/tools/kernel_resources.sh portrayer_amd/csrc/pt_render_m3.o [outdir]
# Also leaves the disassembly in <outdir>/<name>.s (default /tmp/isa).
OBJ=$1; OUT=${2:-/tmp/isa}; LLVM=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"; B=$(basename "$OBJ" .o); W=$(mktemp -d)
cp "$OBJ" "$W/$B.o"
( cd "$W" && $LLVM/llvm-objdump --offloading "$B.o" > /dev/null 2>&1 )
CO=$(ls "$W"/*gfx950* 2>/dev/null | head -1)
[ -z "$CO" ] && { echo "no gfx950 code object in $OBJ"; exit 1; }
cp "$CO" "$OUT/$B.co"
$LLVM/llvm-objdump -d "$OUT/$B.co" > "$OUT/$B.s"
$LLVM/llvm-readelf --notes "$OUT/$B.co" | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    blk='.agpr_count:'+blk
    g=lambda k:(re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    print('%-58s vgpr %3s agpr %3s sgpr %3s scratch %5s B  sgpr_spill %3s vgpr_spill %3s' % (g('name')[:58],g('vgpr_count'),g('agpr_count'),g('sgpr_count'),g('private_segment_fixed_size'),g('sgpr_spill_count'),g('vgpr_spill_count')))
"
rm -rf "$W"
