cp portrayer_amd/libportrayer_hip.so /tmp/keep.so; cp build/variants/kdpk/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
python -m pytest tests -m gpu -x -q -k "kd or KD" > gpurun_out/c64_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c64_pytest.log
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
for i in 1 2; do
bash profiles/variants.sh "kdpk" "big-scene --traversal kd"
done > gpurun_out/c64_ab.log 2>&1
