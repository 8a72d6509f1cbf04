mkdir -p gpurun_out/j9
python -m pytest tests/test_conv_gpu.py::test_conv_step_b256_vs_the_reference tests/test_hip_parity.py::test_large_component_step_vs_the_reference tests/test_distributed_gpu.py::test_two_rank_conv_step_equals_the_oracle -m gpu -q -s 2>&1 | tail -40 > gpurun_out/j9/pytest.txt
python bench.py --config conv --no-cpu-baseline --force-dp > gpurun_out/j9/conv_dp.json 2> gpurun_out/j9/conv_dp.err
python bench.py --config conv --no-cpu-baseline > gpurun_out/j9/conv.json 2> gpurun_out/j9/conv.err
python tools/bench_batch.py > gpurun_out/j9/batch.txt 2>&1
