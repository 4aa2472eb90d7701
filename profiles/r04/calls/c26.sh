# round 4, call 26: 6 waves per SIMD as the densest instantiation of the mesh-free flat_scene / hierarchical kernels (in-tree build): the suite, a fuzz run of those
# kernels at one pixel per wavefront, the default bench line (cold and warm preparation), the profile of the new headline kernel
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 -x > gpurun_out/c26_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c26_pytest.log
FUZZ_MODES=flat,hier timeout 600 python3 tests/fuzz_gpu_parity.py 76000 30 96 64 64 > gpurun_out/c26_fuzz64.log 2>&1
FUZZ_MODES=flat,hier timeout 600 python3 tests/fuzz_gpu_parity.py 77000 40 > gpurun_out/c26_fuzz2.log 2>&1
timeout 900 python3 bench.py > gpurun_out/c26_bench.json 2> gpurun_out/c26_bench.err; echo "rc $?" >> gpurun_out/c26_bench.err
bash profiles/run_profile.sh r04_bigscene --workload big-scene > gpurun_out/c26_prof1.log 2>&1
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c26_prof2.log 2>&1
