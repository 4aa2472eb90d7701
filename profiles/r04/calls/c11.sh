# round 4, call 11: in-tree build = the k-d walk with mode 9 (plain meshes, no KDMesh walker) and the new wave counts: A/B of waves for cows / mirror in the k-d semantics,
# the whole suite (with the new headline-frame tests), k-d fuzz
run() { name=$1; wl=$2; shift; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl --traversal kd 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']
print('%-34s %9.1f Mray/s %8.2f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel']))"
}
run "cows kd default (4 waves)" cows X=1 >> gpurun_out/c11_ab.txt
run "cows kd 3 waves" cows PORTRAYER_KD_WAVES=3 >> gpurun_out/c11_ab.txt
run "mirror kd default (chain 4 waves)" mirror X=1 >> gpurun_out/c11_ab.txt
run "mirror kd chain 3 waves" mirror PORTRAYER_CHAIN_WAVES=3 >> gpurun_out/c11_ab.txt
run "big-scene kd default (5 waves)" big-scene X=1 >> gpurun_out/c11_ab.txt
run "big-soup kd" big-soup X=1 >> gpurun_out/c11_ab.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c11_pytest.log
FUZZ_MODES=kd timeout 600 python3 tests/fuzz_gpu_parity.py 75000 40 > gpurun_out/c11_fuzz.log 2>&1
