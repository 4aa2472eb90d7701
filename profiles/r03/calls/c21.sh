# round 3, call 21: big-scene's kernel time by taking things away, part 2: no shadow walks (nothing ever in the way), no walks at all
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for v in a5 a6; do
  cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
  echo "== build $v (-DPT_ABLATE=${v#a}: 5 no shadow walks, 6 no walks at all)" >> gpurun_out/c21_slope.log
  timeout 300 python3 profiles/light_slope.py flat >> gpurun_out/c21_slope.log 2>&1
done
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
