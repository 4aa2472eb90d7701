# round 4, call 37: the whole suite + smoke + the default bench line on the tree with the texture maps before the state machine, the octant-sorted mesh steps and the
# edge-form triangle records; then the round's second profile set: the workloads whose kernels changed (mirror, big-soup x64, aquarium) and big-scene again
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c37_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c37_pytest.log
tail -3 gpurun_out/c37_pytest.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c37_smoke.log 2>&1; tail -1 gpurun_out/c37_smoke.log
timeout 900 python3 bench.py > gpurun_out/c37_bench.json 2> gpurun_out/c37_bench.err; echo "rc $?" >> gpurun_out/c37_bench.err
bash profiles/run_profile.sh r04_mirror --workload mirror > gpurun_out/c37_prof1.log 2>&1
bash profiles/run_profile.sh r04_soup64 --workload big-soup --samples 64 > gpurun_out/c37_prof2.log 2>&1
bash profiles/run_profile.sh r04_aquarium --workload aquarium > gpurun_out/c37_prof3.log 2>&1
bash profiles/workloads.sh > gpurun_out/c37_workloads.txt 2>&1
cat gpurun_out/c37_workloads.txt
