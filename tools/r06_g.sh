#!/bin/bash
out=gpurun_out/r06g
mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_input_pipeline_gpu.py tests/test_model_api_gpu.py -x -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/summary.txt
tail -n 12 $out/pytest.log
timeout 300 python tools/bench_batch.py 100 112 128 2>&1 | grep 'B=' | tee -a $out/summary.txt
timeout 300 python tools/bench_epoch.py 2>&1 | tail -2 | tee -a $out/summary.txt
