for i in 1 2; do
bash profiles/variants.sh "nopacket packet" "big-scene" "big-scene --traversal hier"
done > gpurun_out/c53_ab.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/c53_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c53_pytest.log
