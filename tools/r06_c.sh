#!/bin/bash
# (historical: ran against commits 1a5dcdb / 405fdfa, whose library still had the MVAE_STEP5 / MVAE_GF / MVAE_L56_GATE switches; kept as the record of the A/B behind DESIGN section 5)
out=gpurun_out/r06c
mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lite_backward or strict or shape_sweep or fused_step" > $out/pytest_parity.log 2>&1
echo "pytest parity rc=$?" | tee -a $out/summary.txt
timeout 900 python -m pytest tests/test_model_api_gpu.py -x -q -m gpu -k "log_likelihood or decoder or loglik" > $out/pytest_ll.log 2>&1
echo "pytest ll rc=$?" | tee -a $out/summary.txt
for rep in 1 2; do
for cfg in "default A=1" "nogate MVAE_L56_GATE=0" "nogf MVAE_GF=0" "neither MVAE_L56_GATE=0,MVAE_GF=0" "five MVAE_STEP5=1"; do
    set -- $cfg
    env $(echo $2 | tr ',' ' ') timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-configs > $out/bench_$1_$rep.json 2> $out/bench_$1_$rep.err
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
MVAE_HIP_LIB=$PWD/mvae_amd/libmvae_hip_timing.so timeout 300 python tools/phase_timing.py > $out/phase.log 2>&1
echo "phase rc=$?" | tee -a $out/summary.txt
python tools/determinism_check.py > $out/determinism.log 2>&1; echo "determinism rc=$?" | tee -a $out/summary.txt
tail -n 5 $out/pytest_parity.log; tail -n 8 $out/pytest_ll.log; tail -n 14 $out/phase.log
