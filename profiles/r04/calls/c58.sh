cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for v in current v_noargs v_nopacket v_o1; do
  if [ $v != current ]; then cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; fi
  python3 profiles/r04/park0_probe3.py $v 2>&1 | tail -2
done > gpurun_out/c58_park0.txt 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
cat gpurun_out/c58_park0.txt
