# round 4, call 61: the wrong counting render of c56 rebuilt on purpose (-DPT_ARGS_AGAIN_EVERYWHERE) under compiler switches that would point at undefined behaviour in the source
# (-fno-strict-aliasing, -fwrapv -fno-delete-null-pointer-checks) or at the scalar-register spill path (-amdgpu-spill-sgpr-to-vgpr=0)
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for v in bad bad_nsa bad_s2m bad_wrapv; do
  cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
  python3 profiles/r04/park0_probe3.py $v 2>&1 | tail -2
done > gpurun_out/c61_park0.txt 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
cat gpurun_out/c61_park0.txt
