#!/bin/bash
# Conv step: the weight planes riding on the latent forward (default) against their own k_split3 launch (MVAE_SPLIT_RIDE=0).
mkdir -p gpurun_out/r06conv; O=gpurun_out/r06conv; : > $O/summary.txt
echo skip-pytest >> $O/summary.txt
for rep in 1 2 3 4 5 6; do
  for ride in 1 0; do
    MVAE_SPLIT_RIDE=$ride timeout 300 python bench.py --config conv --steps 200 --warmup 50 --no-cpu-baseline --no-extra-configs > $O/c${ride}_$rep.json 2> $O/c${ride}_$rep.err
    python - $O/c${ride}_$rep.json $ride $rep <<'PY' >> gpurun_out/r06conv/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ride", sys.argv[2], "rep", sys.argv[3], round(d["value"], 1), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
cat $O/summary.txt
