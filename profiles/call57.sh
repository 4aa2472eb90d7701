for i in 1 2; do
bash profiles/variants.sh "pkmesh" "cows" "big-soup --samples 64" "big-soup" "big-mesh" "big-mesh --samples 64"
done > gpurun_out/c57_ab.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so; cp build/variants/pkmesh/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
python -m pytest tests -m gpu -x -q > gpurun_out/c57_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c57_pytest.log
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
