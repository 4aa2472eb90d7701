export PORTRAYER_WAVES=3
bash profiles/run_profile.sh r02_bigscene_w3 --workload big-scene > /dev/null 2>&1
bash profiles/run_profile.sh r02_soup64_w3 --workload big-soup --samples 64 > /dev/null 2>&1
for t in r02_bigscene_w3 r02_soup64_w3; do python3 profiles/digest.py $t; done > gpurun_out/c70_digest.log 2>&1
