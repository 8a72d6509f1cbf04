#!/bin/bash
# (historical: ran against commits 1a5dcdb / 405fdfa, whose library still had the MVAE_STEP5 / MVAE_GF / MVAE_L56_GATE switches; kept as the record of the A/B behind DESIGN section 5)
# round 6: the four-launch step (k_bwd56) against the five-launch lite step (MVAE_STEP5=1), with and without g's
# fragment-order copy (MVAE_GF=1); parity tests first, then interleaved 2000-step bench runs.
out=gpurun_out/r06ab
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lite_backward or fused_step or kernel_path or wide_hidden" > $out/pytest_default.log 2>&1
echo "pytest default rc=$?" | tee -a $out/summary.txt
MVAE_GF=1 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lite_backward or fused_step" > $out/pytest_gf.log 2>&1
echo "pytest gf rc=$?" | tee -a $out/summary.txt
for rep in 1 2; do
  for cfg in "five MVAE_STEP5=1" "four MVAE_STEP5=0" "fourgf MVAE_GF=1"; do
    set -- $cfg
    env $2 timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-configs > $out/bench_$1_$rep.json 2> $out/bench_$1_$rep.err
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.loads(open("$out/bench_$1_$rep.json").read().strip().splitlines()[-1])
    pk=d["roofline"].get("per_kernel")
    print("$1 rep$rep", round(d["value"]), "steps/s", round(d["ms_per_step"]*1e3,2), "us", {k: round(v["ms"]*1e3,2) for k,v in pk.items()})
except Exception as e:
    print("$1 rep$rep failed", e)
PY
  done
done
for m in e6; do
  for cfg in "five MVAE_STEP5=1" "four MVAE_STEP5=0"; do
    set -- $cfg
    env $2 timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-configs --model $m --fixed-curvature > $out/bench_${m}_$1.json 2> $out/bench_${m}_$1.err
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.loads(open("$out/bench_${m}_$1.json").read().strip().splitlines()[-1])
    print("$m $1", round(d["value"]), "steps/s", round(d["ms_per_step"]*1e3,2), "us")
except Exception as e:
    print("$m $1 failed", e)
PY
  done
done
python tools/determinism_check.py > $out/determinism.log 2>&1; echo "determinism rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_default.log $out/pytest_gf.log $out/determinism.log
