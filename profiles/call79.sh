bash profiles/examples_vs_goldens.sh > gpurun_out/c79_examples.log 2>&1
( cd tests/golden; for b in primitives robot-alarm-clock big-scene single-triangle macho-cows hier instance antialiasing fish simple nonhier nonhier2 four-shapes graphics-temple graphics-poster simple-cows; do
  SAMPLES=4 timeout 300 ../../examples/bin/$b > /dev/null 2> err.txt; echo "$b rc $? $(ls *.png 2>/dev/null | tr '\n' ' ')"; rm -f *.png err.txt; done ) > gpurun_out/c79_run.log 2>&1
timeout 1200 python tests/fuzz_gpu_parity.py 150000 300 200 150 2 > gpurun_out/c79_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 151000 100 96 72 16 >> gpurun_out/c79_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 152000 60 50 38 100 >> gpurun_out/c79_fuzz.log 2>&1
