# round 3, call 20: big-scene's kernel time by taking things away: 0..3 lights, and the light term ablated (no specular / no term / the device library's pow)
timeout 300 python3 profiles/light_slope.py flat > gpurun_out/c20_slope.log 2>&1
timeout 300 python3 profiles/light_slope.py hier >> gpurun_out/c20_slope.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for v in a1 a2 a3; do
  cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
  echo "== build $v (-DPT_ABLATE=${v#a}: 1 no specular term, 2 no light term, 3 the device library's pow)" >> gpurun_out/c20_slope.log
  timeout 300 python3 profiles/light_slope.py flat >> gpurun_out/c20_slope.log 2>&1
done
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
