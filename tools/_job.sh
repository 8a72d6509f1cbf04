mkdir -p gpurun_out/j15
python -m pytest tests/test_ops_gpu.py tests/test_model_api_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/j15/pytest.txt
