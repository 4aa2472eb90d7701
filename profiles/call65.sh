( export PORTRAYER_LDS_BUDGET_KB=39; for i in 1 2; do bash profiles/variants.sh "w4" "big-scene" "mirror" "big-soup --samples 64" "aquarium" "cows"; done ) > gpurun_out/c65_w4.log 2>&1
( export PORTRAYER_LDS_BUDGET_KB=79; bash profiles/variants.sh "w2" "big-scene" "mirror" "big-soup --samples 64" "aquarium" "cows" ) > gpurun_out/c65_w2.log 2>&1
