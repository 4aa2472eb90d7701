# round 3, call 19b: PC sampling of the headline kernel (rocprofv3 beta feature): what the box supports, then the first configuration that works
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/c19
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 60 rocprofv3-avail list --pc-sampling > $OUT/avail.log 2>&1
timeout 60 rocprofv3-avail info --pc-sampling >> $OUT/avail.log 2>&1
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload big-scene"
for cfg in "stochastic cycles 1048576" "stochastic cycles 262144" "stochastic cycles 4096" "host_trap time 1000" "host_trap time 10000" "host_trap time 1"; do
  set -- $cfg
  D=$OUT/pcs_$1_$3
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d $D -- python3 $ROOTDIR/bench.py $ARGS > $D.log 2>&1; RC=$?
  echo "rc $RC" >> $D.log
  if [ $RC -eq 0 ] && [ -n "$(find $D -name '*pc_sampling*.csv' 2>/dev/null)" ]; then break; fi
done
for f in $(find $OUT -name "*pc_sampling*.csv"); do
  python3 - "$f" <<'PY' >> $OUT/hist.log 2>&1
import csv, sys, collections
f = sys.argv[1]
rows = csv.reader(open(f))
head = next(rows)
print(f, head)
n = 0
c = collections.Counter()
first = []
for r in rows:
    n += 1
    if len(first) < 5: first.append(r)
    d = dict(zip(head, r))
    key = (d.get('Instruction', ''), d.get('Instruction_Comment', ''), d.get('Code_Object_Id', ''), d.get('Code_Object_Offset', ''))
    c[key] += 1
print(n, 'samples'); print(first)
with open(f + '.hist', 'w') as o:
    for k, v in c.most_common(): o.write('%d\t%s\n' % (v, '\t'.join(k)))
PY
  SZ=$(stat -c %s "$f"); if [ "$SZ" -gt 20000000 ]; then rm "$f"; fi
done
find $OUT -type f > $OUT/files.txt; du -sh $OUT >> $OUT/files.txt
