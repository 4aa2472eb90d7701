#!/bin/bash
# One PMC pass over bench.py; prints the mean per render-kernel launch of each counter.
# usage: bash profiles/pmc_quick.sh "<counters>" [bench args]
CTRS=$1; shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmcq_$$
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOTDIR/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections,re
f=glob.glob('$OUT/*/*_counter_collection.csv')
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if re.search(r'pt_render(_simple)?_kernel<\d+, false,', r['Kernel_Name']):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('%-28s %.4g' % (k, sum(v)/len(v)))
PY
tail -1 $OUT/log.txt | cut -c1-160
