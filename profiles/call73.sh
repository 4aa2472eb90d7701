for i in 1 2; do
bash profiles/variants.sh "base" "big-scene" "big-soup --samples 64" "big-mesh" "big-scene --width 3840 --height 2160 --samples 256 --steps 2"
done > gpurun_out/c73_ab.log 2>&1
( bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload big-scene; bash profiles/pmc_quick.sh "FETCH_SIZE" --no-extras --workload big-scene; bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload big-soup --samples 64 ) > gpurun_out/c73_pmc.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/c73_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c73_pytest.log
