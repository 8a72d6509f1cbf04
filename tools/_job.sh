mkdir -p gpurun_out/j8
python -m pytest tests/test_bench_contract_gpu.py tests/test_distributed_gpu.py tests/test_conv_gpu.py -m gpu -q -x --deselect tests/test_conv_gpu.py::test_conv_step_at_the_baseline_batch_256 2>&1 | tail -40 > gpurun_out/j8/pytest.txt
python bench.py --no-cpu-baseline --no-extra-configs --force-dp > gpurun_out/j8/dp.json 2> gpurun_out/j8/dp.err
MVAE_DP_OVERLAP=0 python bench.py --no-cpu-baseline --no-extra-configs --force-dp > gpurun_out/j8/dp_noov.json 2> gpurun_out/j8/dp_noov.err
