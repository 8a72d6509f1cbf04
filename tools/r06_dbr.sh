#!/bin/bash
out=gpurun_out/r06dbr
mkdir -p $out
timeout 600 python -m pytest tests/test_model_api_gpu.py -x -q -m gpu -k "log_likelihood or decoder or loglik" > $out/pytest_ll.log 2>&1
echo "pytest ll (32-row kernel) rc=$?" | tee -a $out/summary.txt
for z in 6 48; do
for v in 1 0 1 0; do
  MVAE_DBR32=$v timeout 120 python tools/bench_decode_bce.py $z 2>&1 | tail -1 | sed "s/^/dbr32=$v z=$z /" | tee -a $out/summary.txt
done
done
for v in 1 0; do MVAE_DBR32=$v timeout 200 python tools/bench_ll.py 2>&1 | tail -3 | sed "s/^/dbr32=$v /" | tee -a $out/summary.txt; done
tail -n 5 $out/pytest_ll.log
