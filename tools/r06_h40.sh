#!/bin/bash
mkdir -p gpurun_out/r06h40; O=gpurun_out/r06h40; : > $O/summary.txt
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest parity rc=$?" >> $O/summary.txt; tail -n 3 $O/pytest.log >> $O/summary.txt
for m in h40 s40 p40 h16,s16 h2,s2,e2; do
  for nl in 0 1; do
    MVAE_NO_LITE=$nl timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --model $m --steps 500 --warmup 50 > $O/${m}_$nl.json 2> $O/${m}_$nl.err
    python - $O/${m}_$nl.json $m $nl <<'PY' >> gpurun_out/r06h40/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "NO_LITE", sys.argv[3], round(d["value"]), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us", {k: round(v * 1e3, 2) for k, v in d["roofline"].get("kernel_ms", {}).items()})
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
cat $O/summary.txt
