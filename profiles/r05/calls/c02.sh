# round 5, call 2: which counters of the wrong render differ, and from run to run; then the multi-rank tests with this round's pt_node changes
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
O=portrayer_amd/csrc
objs="$O/pt_api.o $O/pt_build.o $O/pt_node.o"; for m in 1 3 4 5 6 7 8 9; do objs="$objs $O/pt_render_m$m.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs build/diag/bad.o -o portrayer_amd/libportrayer_hip.so -ldl
timeout 300 python3 profiles/r05/park0_stats.py bad > gpurun_out/c02_stats.txt 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
cat gpurun_out/c02_stats.txt
timeout 900 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/c02_tests.txt 2>&1; tail -3 gpurun_out/c02_tests.txt
