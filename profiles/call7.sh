rm -f gpurun_out/ulp_report.txt
PT_ULP_LOG=gpurun_out/ulp_report.txt python -m pytest tests -m gpu -x -q -s > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c7_pytest.log
python -c "
import sys; sys.path.insert(0,'.')
from portrayer_amd import _hip as H
c = H.Context(0); print('copy bandwidth GB/s', c.copy_bandwidth(1<<30, 5))" > gpurun_out/c7_bw.log 2>&1
