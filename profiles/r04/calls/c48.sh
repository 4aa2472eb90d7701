# round 4, call 48: the final tree: whole suite, smoke, the default bench line, all workloads, and the round's profile sets again (every kernel was rebuilt since c21 / c37)
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c48_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c48_pytest.log
tail -3 gpurun_out/c48_pytest.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c48_smoke.log 2>&1; tail -1 gpurun_out/c48_smoke.log
timeout 900 python3 bench.py > gpurun_out/c48_bench.json 2> gpurun_out/c48_bench.err; echo "rc $?" >> gpurun_out/c48_bench.err
bash profiles/run_profile.sh r04_bigscene --workload big-scene > gpurun_out/c48_prof1.log 2>&1
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c48_prof2.log 2>&1
bash profiles/run_profile.sh r04_kd --workload big-scene --traversal kd > gpurun_out/c48_prof3.log 2>&1
bash profiles/run_profile.sh r04_mirror --workload mirror > gpurun_out/c48_prof4.log 2>&1
bash profiles/run_profile.sh r04_soup64 --workload big-soup --samples 64 > gpurun_out/c48_prof5.log 2>&1
bash profiles/run_profile.sh r04_aquarium --workload aquarium > gpurun_out/c48_prof6.log 2>&1
bash profiles/workloads.sh > gpurun_out/c48_workloads.txt 2>&1
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload water-glass 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('--workload water-glass %9.1f Mray/s %8.2f ms/frame' % (d['value'], d['ms_per_step']))" >> gpurun_out/c48_workloads.txt
cat gpurun_out/c48_workloads.txt
