# round 4, call 51: hierarchical semantics - identity levels skipped on the way UP (hit point and normal) too; then the final tree: whole suite, smoke, default bench line, the hierarchical profile set, all workloads
FUZZ_MODES=hier timeout 900 python3 tests/fuzz_gpu_parity.py 91000 60 > gpurun_out/c51_fuzz.log 2>&1; tail -1 gpurun_out/c51_fuzz.log
bash profiles/variants.sh "alllevels" "big-scene --traversal hier" "big-scene --traversal hier" "mirror --traversal hier" "cows --traversal hier" "aquarium --traversal hier" "water-glass --traversal hier" > gpurun_out/c51_variants.txt 2>&1
cat gpurun_out/c51_variants.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c51_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c51_pytest.log
grep -n "passed\|failed" gpurun_out/c51_pytest.log | tail -1
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c51_smoke.log 2>&1; tail -1 gpurun_out/c51_smoke.log
timeout 900 python3 bench.py > gpurun_out/c51_bench.json 2> gpurun_out/c51_bench.err; echo "rc $?" >> gpurun_out/c51_bench.err
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c51_prof2.log 2>&1
bash profiles/workloads.sh > gpurun_out/c51_workloads.txt 2>&1
cat gpurun_out/c51_workloads.txt
