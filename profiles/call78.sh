python -m pytest tests -m gpu -x -q > gpurun_out/c78_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c78_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c78_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/c78_smoke.log
python bench.py --gpus 2 --backend gloo --same-device --check --steps 2 --warmup 1 > gpurun_out/c78_bench2.log 2>&1; echo "rc $?" >> gpurun_out/c78_bench2.log
bash profiles/run_profile.sh r02_bigscene --workload big-scene > /dev/null 2>&1
bash profiles/run_profile.sh r02_hier --workload big-scene --traversal hier > /dev/null 2>&1
bash profiles/run_profile.sh r02_kd --workload big-scene --traversal kd > /dev/null 2>&1
bash profiles/run_profile.sh r02_soup64 --workload big-soup --samples 64 > /dev/null 2>&1
bash profiles/run_profile.sh r02_mirror --workload mirror > /dev/null 2>&1
bash profiles/run_profile.sh r02_aquarium --workload aquarium > /dev/null 2>&1
for t in r02_bigscene r02_hier r02_kd r02_soup64 r02_mirror r02_aquarium; do python3 profiles/digest.py $t; done > gpurun_out/c78_digest.log 2>&1
bash profiles/workloads.sh --no-extras > gpurun_out/c78_workloads.log 2>&1
bash profiles/diag.sh "--workload big-scene" "--workload big-soup --samples 64" "--workload mirror" "--workload aquarium" "--workload cows" "--workload big-scene --traversal kd" > gpurun_out/c78_diag.log 2>&1
bash profiles/timeline.sh "--workload big-scene" "--workload big-soup --samples 64" "--workload mirror" "--workload aquarium" "--workload cows" "--workload big-scene --traversal kd" > gpurun_out/c78_timeline.log 2>&1
bash profiles/cycles.sh "--workload big-scene" "--workload aquarium" "--workload mirror" "--workload big-soup --samples 64" > gpurun_out/c78_cycles.log 2>&1
for wl in "big-scene --width 3840 --height 2160 --samples 256" "big-scene --width 3840 --height 2160 --samples 256 --share 8" "big-scene --share 2" "big-scene --share 4" "big-scene --share 8" "cows --traversal hier" "cows --traversal kd" "mirror --traversal kd" "big-soup --samples 64" "big-soup --samples 64 --traversal hier" "aquarium --traversal hier" "aquarium --samples 64 --steps 2" "big-mesh --samples 64"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-70s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done > gpurun_out/c78_more.log 2>&1
PORTRAYER_BUILD=host python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload big-soup --samples 64 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/c78_more.log
python bench.py --steps 10 --warmup 3 > gpurun_out/c78_bench.log 2>&1; echo "rc $?" >> gpurun_out/c78_bench.log
