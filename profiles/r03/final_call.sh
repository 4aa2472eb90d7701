# (round 3, call c39) The round's final profile set on the default build (kernel trace + PMC passes per workload), the default bench line, the workload table, the suite
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/c39_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c39_pytest.log
bash profiles/run_profile.sh r03_bigscene --workload big-scene > gpurun_out/c39_prof1.log 2>&1
bash profiles/run_profile.sh r03_hier --workload big-scene --traversal hier > gpurun_out/c39_prof2.log 2>&1
bash profiles/run_profile.sh r03_kd --workload big-scene --traversal kd > gpurun_out/c39_prof3.log 2>&1
bash profiles/run_profile.sh r03_soup64 --workload big-soup --samples 64 > gpurun_out/c39_prof4.log 2>&1
bash profiles/run_profile.sh r03_mirror --workload mirror > gpurun_out/c39_prof5.log 2>&1
bash profiles/run_profile.sh r03_aquarium --workload aquarium > gpurun_out/c39_prof6.log 2>&1
python3 bench.py > gpurun_out/c39_bench.json 2> gpurun_out/c39_bench.err
bash profiles/workloads.sh > gpurun_out/c39_workloads.log 2>&1
for wl in "big-soup --samples 64" "big-mesh --samples 64" "big-scene --share 8" "big-scene --width 3840 --height 2160 --samples 256" "cows --traversal hier" "aquarium --traversal hier" "aquarium --samples 64" "mirror --traversal kd" "water-glass" "water-glass --traversal hier"; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-60s %9.1f Mray/s %9.2f ms/frame  %s' % ('$wl', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))" >> gpurun_out/c39_workloads.log 2>&1
done
