#!/bin/bash
mkdir -p gpurun_out/r06k; O=gpurun_out/r06k
timeout 1200 python -m pytest tests/test_input_pipeline_gpu.py tests/test_model_api_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/summary.txt
tail -4 $O/pytest.log >> $O/summary.txt
for i in 1 2 3; do timeout 200 python tools/bench_ll.py >> $O/summary.txt 2>&1; done
cat $O/summary.txt
