bash profiles/run_profile.sh aq --workload aquarium > /dev/null 2>&1
bash profiles/run_profile.sh soup --workload big-soup > /dev/null 2>&1
bash profiles/run_profile.sh kd --workload big-scene --traversal kd > /dev/null 2>&1
for t in aq soup kd; do python3 profiles/digest.py $t; done > gpurun_out/c4_digest.log 2>&1
(echo notex; PORTRAYER_NO_TEX=1 python3 bench.py --no-cpu-baseline --no-extras --workload aquarium --steps 2 | tail -1 | cut -c1-200
echo 2blocks; PORTRAYER_BLOCKS_PER_CU=2 python3 bench.py --no-cpu-baseline --no-extras --workload aquarium --steps 2 | tail -1 | cut -c1-200
echo kd3waves; PORTRAYER_WAVES=3 python3 bench.py --no-cpu-baseline --no-extras --workload big-scene --traversal kd --steps 2 | tail -1 | cut -c1-200
echo soup3waves; PORTRAYER_WAVES=3 python3 bench.py --no-cpu-baseline --no-extras --workload big-soup --steps 2 | tail -1 | cut -c1-200
echo soup-ldsstack12; PORTRAYER_LDS_STACK=12 python3 bench.py --no-cpu-baseline --no-extras --workload big-soup --steps 2 | tail -1 | cut -c1-200
echo soup-host-tree; PORTRAYER_BUILD=host python3 bench.py --no-cpu-baseline --no-extras --workload big-soup --steps 2 | tail -1 | cut -c1-200
) >> gpurun_out/c4_digest.log 2>&1
