for i in 1 2; do
bash profiles/variants.sh "packet" "big-scene" "big-scene --share 8" "big-scene --width 960 --height 540 --samples 16"
done > gpurun_out/c51_ab.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so; cp build/variants/packet/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
python -m pytest tests -m gpu -x -q -k "big_scene or big-scene or flat" > gpurun_out/c51_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c51_pytest.log
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
