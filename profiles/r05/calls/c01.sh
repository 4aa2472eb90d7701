# round 5, call 1: the wrong counting render (notes r04 section 11) under ten diagnostic builds of the reproducer, to find what it depends on
bash profiles/r05/diag_matrix.sh gpurun_out/c01_diag.txt bad bad_w0 bad_2w bad_nopk bad_nosched bad_noipra bad_powinl bad_O2 bad_nomlicm bad_s2m
# ... and the tests this round's first changes touch (plain instantiation at C1 / C2 size, pt_node's error paths)
timeout 900 python3 -m pytest tests/test_gpu_render_parity.py -k "test_example_matches_oracle" tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/c01_tests.txt 2>&1; tail -3 gpurun_out/c01_tests.txt
