# round 4, call 10: A/B of the k-d walk on big-scene (kernel ms, VALU / SALU instructions per frame): axis-specialised split arithmetic against run-time selects,
# culls on / off (nodes, references), 3 / 4 / 5 waves per SIMD; mirror and cows with the new build
run() { # name, env...
  name=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload ${WL:-big-scene} --traversal kd 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']
print('%-34s %9.1f Mray/s %8.2f ms  nodes/ray %.2f prim/ray %.2f' % ('$name', d['value'], d['ms_per_step'], r['inner_nodes'], r['primitive_tests']))"
}
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/kdsel/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
run "axis by select" X=1 >> gpurun_out/c10_ab.txt
cp build/variants/kdw/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
run "axis-specialised (new default)" X=1 >> gpurun_out/c10_ab.txt
run "  no node cull" PORTRAYER_KD_CULL=2 >> gpurun_out/c10_ab.txt
run "  no reference cull" PORTRAYER_KD_CULL=1 >> gpurun_out/c10_ab.txt
run "  no cull at all" PORTRAYER_KD_CULL=0 >> gpurun_out/c10_ab.txt
run "  3 waves" PORTRAYER_KD_WAVES=3 >> gpurun_out/c10_ab.txt
run "  5 waves" PORTRAYER_KD_WAVES=5 >> gpurun_out/c10_ab.txt
WL=mirror run "mirror" X=1 >> gpurun_out/c10_ab.txt
WL=cows run "cows" X=1 >> gpurun_out/c10_ab.txt
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
bash profiles/pmc_quick.sh "$C1" --no-extras --workload big-scene --traversal kd > gpurun_out/c10_pmc_1.txt 2>&1
FUZZ_MODES=kd timeout 300 python3 tests/fuzz_gpu_parity.py 74000 20 > gpurun_out/c10_fuzz.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
