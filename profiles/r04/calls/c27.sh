# round 4, call 27: what the dielectric scene's 148 GB per frame are made of: the same frame through the UNTEXTURED kernel (PORTRAYER_NO_TEX=1: 64 B of scratch a lane
# instead of 528; wrong picture, timing and traffic only), with every parked frame in HBM (PORTRAYER_PARK=0), and the HBM-side bytes of each
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:70]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload aquarium"
run "aquarium default" X=1 $B >> gpurun_out/c27_aq.txt
run "aquarium NO_TEX" PORTRAYER_NO_TEX=1 $B >> gpurun_out/c27_aq.txt
run "aquarium PARK=0" PORTRAYER_PARK=0 $B >> gpurun_out/c27_aq.txt
run "aquarium hier" X=1 $B --traversal hier >> gpurun_out/c27_aq.txt
for v in "default:X=1" "notex:PORTRAYER_NO_TEX=1"; do n=${v%%:*}; e=${v#*:}
  env $e bash profiles/pmc_quick.sh "FETCH_SIZE" --no-extras --workload aquarium > gpurun_out/c27_pmc_fetch_$n.txt 2>&1
  env $e bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload aquarium > gpurun_out/c27_pmc_write_$n.txt 2>&1
  env $e bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" --no-extras --workload aquarium > gpurun_out/c27_pmc_sq_$n.txt 2>&1
done
