#!/bin/bash
out=gpurun_out/r06e
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lite_backward or strict or shape_sweep or fused_step or wide_hidden or other_batch" > $out/pytest_parity.log 2>&1
echo "pytest parity rc=$?" | tee -a $out/summary.txt
for rep in 1 2 3; do
for cfg in "early A=1" "late MVAE_HIP_LIB=$PWD/mvae_amd/_variants/libmvae_hip_f23late.so"; do
    set -- $cfg
    env $2 timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-configs > $out/bench_$1_$rep.json 2> $out/bench_$1_$rep.err
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.loads(open("$out/bench_$1_$rep.json").read().strip().splitlines()[-1])
    print("$1 rep$rep", round(d["value"]), "steps/s", round(d["ms_per_step"]*1e3,2), "us", {k: round(v["ms"]*1e3,2) for k,v in d["roofline"]["per_kernel"].items()})
except Exception as e:
    print("$1 rep$rep failed", e)
PY
done
done
MVAE_HIP_LIB=$PWD/mvae_amd/libmvae_hip_timing.so timeout 300 python tools/phase_timing.py 2>&1 | grep 'fwd23\|dec1_fwd'
python tools/determinism_check.py > $out/determinism.log 2>&1; echo "determinism rc=$?" | tee -a $out/summary.txt
tail -n 3 $out/pytest_parity.log
