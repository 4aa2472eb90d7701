python -m pytest tests -m gpu -x -q > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c5_pytest.log
bash profiles/sweep_waves.sh > gpurun_out/c5_sweep.log 2>&1
bash profiles/run_profile.sh aq --workload aquarium > /dev/null 2>&1
python3 profiles/digest.py aq >> gpurun_out/c5_sweep.log 2>&1
