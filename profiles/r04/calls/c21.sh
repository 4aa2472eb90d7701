# round 4, call 21: the whole suite on the in-tree build (watchdogs in the walks, pixel-major chunk sums, chunk-sum loops bounded), the default bench line,
# and the round's first profile set: big-scene in the three semantics (summaries: python3 profiles/summarise.py <tag> r04 <key>, run afterwards where the repo is)
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c21_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c21_pytest.log
timeout 900 python3 bench.py > gpurun_out/c21_bench.json 2> gpurun_out/c21_bench.err; echo "rc $?" >> gpurun_out/c21_bench.err
bash profiles/run_profile.sh r04_bigscene --workload big-scene > gpurun_out/c21_prof1.log 2>&1
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c21_prof2.log 2>&1
bash profiles/run_profile.sh r04_kd --workload big-scene --traversal kd > gpurun_out/c21_prof3.log 2>&1
