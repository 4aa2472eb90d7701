for i in 1 2; do bash profiles/variants.sh "tmc" "big-scene" "big-scene --traversal hier"; done > gpurun_out/c80_ab.log 2>&1
