#!/bin/bash
out=gpurun_out/r06d
mkdir -p $out
for rep in 1 2; do
for cfg in "base A=1" "lds90k MVAE_L56_LDS=92160" "lds60k MVAE_L56_LDS=61440"; do
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
true
