bash profiles/run_profile.sh aq --workload aquarium > /dev/null 2>&1
bash profiles/run_profile.sh soup --workload big-soup > /dev/null 2>&1
bash profiles/run_profile.sh bs --workload big-scene > /dev/null 2>&1
bash profiles/run_profile.sh kd --workload big-scene --traversal kd > /dev/null 2>&1
for t in aq soup bs kd; do python3 profiles/digest.py $t; done > gpurun_out/c3_digest.log 2>&1
du -sh gpurun_out/prof_* >> gpurun_out/c3_digest.log
