# round 4, call 51: the final tree once more: whole suite, smoke, default bench line, the hierarchical profile set (its leaf test changed in c50), all workloads
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c51_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c51_pytest.log
grep -n "passed\|failed" gpurun_out/c51_pytest.log | tail -1
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c51_smoke.log 2>&1; tail -1 gpurun_out/c51_smoke.log
timeout 900 python3 bench.py > gpurun_out/c51_bench.json 2> gpurun_out/c51_bench.err; echo "rc $?" >> gpurun_out/c51_bench.err
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c51_prof2.log 2>&1
bash profiles/workloads.sh > gpurun_out/c51_workloads.txt 2>&1
cat gpurun_out/c51_workloads.txt
