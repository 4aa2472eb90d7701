for i in 1 2; do
bash profiles/variants.sh "pk4" "mirror" "cows" "aquarium" "big-soup --samples 64" "big-soup" "big-mesh" "mirror --traversal hier"
done > gpurun_out/c61_ab.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so; cp build/variants/pk4/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
python -m pytest tests -m gpu -x -q > gpurun_out/c61_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c61_pytest.log
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
