for cfg in "PORTRAYER_WAVES=4" "PORTRAYER_WAVES=3" "PORTRAYER_FINE_QUEUES=0" "PORTRAYER_PARK=0" "PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16" "PORTRAYER_COLLAPSE=area" "PORTRAYER_LANE_CHUNKS=1"; do
  ( for kv in $cfg; do export $kv; done; python -m pytest tests -m gpu -q -k "not stack_overflow_is_reported and not scheduling_and_tree_shape and not traversal_stack_beyond_lds" 2>&1 | grep -E "passed|failed" | sed "s/^/$cfg: /" )
done > gpurun_out/c87.log 2>&1
