# round 4, call 7: whole suite again (fork / join queue moved into the wavefront's own rows in the k-d semantics; lane rows clamped)
timeout 1500 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/c07_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c07_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c07_smoke.log 2>&1; echo "rc $?" >> gpurun_out/c07_smoke.log
