bash tools/r05_trace.sh r05t4 "one:no_overlap=1" cv "nn_ball_listed_items iteration_sums"
bash tools/r05_quick.sh r05k "tests/test_icp_gpu.py tests/test_fused_batch_oracle_gpu.py tests/test_properties_gpu.py" "two:;one:no_overlap=1"
