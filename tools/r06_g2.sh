#!/bin/bash
# The driver's call (--steps 20 --warmup 5) with the timed region as ONE graph against several shorter graphs.
mkdir -p gpurun_out/r06g2; O=gpurun_out/r06g2; : > $O/summary.txt
for rep in 1 2; do
for gs in one 10 5 4 2 1; do
  if [ $gs = one ]; then e="X=1"; a=""; else e="MVAE_BENCH_ONE_GRAPH_MAX=0"; a="--graph-steps $gs"; fi
  env $e timeout 300 python bench.py --steps 20 --warmup 5 $a --no-cpu-baseline --no-extra-configs > $O/g${gs}_$rep.json 2> $O/g${gs}_$rep.err
  python - $O/g${gs}_$rep.json $gs $rep <<'PY' >> gpurun_out/r06g2/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("graph", sys.argv[2], "rep", sys.argv[3], round(d["value"]), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us", d["config"].get("graph_steps"), d["config"].get("graph_replays"), d["config"]["repeat_ms_per_step"]["all"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done
done
cat $O/summary.txt
