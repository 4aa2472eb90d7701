python -m pytest tests -m gpu -x -q > gpurun_out/c48_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c48_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c48_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/c48_smoke.log
python bench.py --gpus 2 --backend gloo --same-device --check --steps 2 --warmup 1 > gpurun_out/c48_bench2.log 2>&1; echo "rc $?" >> gpurun_out/c48_bench2.log
bash profiles/run_profile.sh r02_bigscene --workload big-scene > /dev/null 2>&1
bash profiles/run_profile.sh r02_hier --workload big-scene --traversal hier > /dev/null 2>&1
bash profiles/run_profile.sh r02_kd --workload big-scene --traversal kd > /dev/null 2>&1
bash profiles/run_profile.sh r02_soup64 --workload big-soup --samples 64 > /dev/null 2>&1
bash profiles/run_profile.sh r02_mirror --workload mirror > /dev/null 2>&1
bash profiles/run_profile.sh r02_aquarium --workload aquarium > /dev/null 2>&1
for t in r02_bigscene r02_hier r02_kd r02_soup64 r02_mirror r02_aquarium; do python3 profiles/digest.py $t; done > gpurun_out/c48_digest.log 2>&1
bash profiles/workloads.sh --no-extras > gpurun_out/c48_workloads.log 2>&1
