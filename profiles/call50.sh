bash profiles/examples_vs_goldens.sh > gpurun_out/c50_examples.log 2>&1
( cd tests/golden; for b in primitives robot-alarm-clock big-scene single-triangle macho-cows hier instance antialiasing fish simple nonhier four-shapes graphics-temple; do
  SAMPLES=4 timeout 300 ../../examples/bin/$b > /dev/null 2> err.txt; echo "$b rc $? $(ls *.png 2>/dev/null | tr '\n' ' ')"; rm -f *.png err.txt; done ) > gpurun_out/c50_run.log 2>&1
