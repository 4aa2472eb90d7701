run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( echo "== base"; run aquarium
  echo "== NO_TEX"; PORTRAYER_NO_TEX=1 run aquarium
  echo "== NO_TEX PARK=1"; PORTRAYER_NO_TEX=1 PORTRAYER_PARK=1 run aquarium
  echo "== NO_TEX PARK=2 (LDS 54K)"; PORTRAYER_NO_TEX=1 PORTRAYER_PARK=2 run aquarium
  echo "== PARK=1 LDS_STACK=8 (over budget?)"; PORTRAYER_PARK=1 PORTRAYER_LDS_STACK=8 run aquarium
) > gpurun_out/c22_time.log 2>&1
( for v in "PORTRAYER_PARK=0" "PORTRAYER_PARK=1" "PORTRAYER_NO_TEX=1"; do
  echo "== $v"
  export $v
  bash profiles/pmc_quick.sh "FETCH_SIZE" --no-extras --workload aquarium
  bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload aquarium
  bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" --no-extras --workload aquarium
  unset PORTRAYER_PARK PORTRAYER_NO_TEX
done ) > gpurun_out/c22_pmc.log 2>&1
cd /tmp; rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/c22_counters.txt 2>&1
