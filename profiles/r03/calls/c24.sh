# round 3, call 24: the next queue index asked for one item ahead; per-item overhead again (no walks at all)
bash profiles/workloads.sh --no-extras > gpurun_out/c24_workloads.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/a6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
echo "== build a6 (-DPT_ABLATE=6: no walks at all)" > gpurun_out/c24_slope.log
timeout 300 python3 profiles/light_slope.py flat >> gpurun_out/c24_slope.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c24_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c24_pytest.log
