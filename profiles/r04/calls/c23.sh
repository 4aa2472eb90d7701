# round 4, call 23: the finishing pass reads the pixel-major chunk sums eight threads per pixel (coalesced), the mesh-free walk without its watchdog:
# the suite, the headline twice, the profile of the headline kernel again
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 -x > gpurun_out/c23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c23_pytest.log
for rep in 1 2; do
python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --workload big-scene 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('flat %9.1f Mray/s %8.3f ms (kernel %.3f)' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> gpurun_out/c23_ab.txt
python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --workload big-scene --traversal hier 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('hier %9.1f Mray/s %8.3f ms (kernel %.3f)' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> gpurun_out/c23_ab.txt
done
bash profiles/run_profile.sh r04_bigscene --workload big-scene > gpurun_out/c23_prof1.log 2>&1
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c23_prof2.log 2>&1
