#!/bin/bash
out=gpurun_out/r06f
mkdir -p $out
for v in 0 1; do
  MVAE_FUSED_ANY_B=$v timeout 600 python tools/bench_batch.py 16 32 64 96 112 192 256 2>&1 | grep 'B=' | sed "s/^/any_b=$v /" | tee -a $out/summary.txt
done
MVAE_FUSED_ANY_B=1 timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu > $out/pytest_parity.log 2>&1
echo "pytest parity (any_b) rc=$?" | tee -a $out/summary.txt
tail -n 6 $out/pytest_parity.log
