# round 4, call 45: the many-triangle scenes at 6 waves per SIMD (80 registers, 132 spilled - some inside the walk) against the shipped 5
bash profiles/variants.sh "mesh6" "big-soup --samples 64" "big-mesh --samples 64" "big-soup --samples 64 --traversal hier" big-soup > gpurun_out/c45_variants.txt 2>&1
cat gpurun_out/c45_variants.txt
