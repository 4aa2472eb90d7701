#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel trace + stats, then PMC passes (separate runs, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (from the repo root on the box): bash profiles/run_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOTDIR/bench.py $ARGS > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOTDIR/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOTDIR/bench.py $ARGS > $OUT/bench_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOTDIR/bench.py $ARGS > $OUT/bench_sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $ROOTDIR/bench.py $ARGS > $OUT/bench_tcc.log 2>&1
find $OUT -name "*.csv" | head -50
