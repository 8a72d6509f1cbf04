#!/bin/bash
# World-1 forced exchange (bench.py --force-dp) on every route: what one GPU can measure of the data-parallel tax.
mkdir -p gpurun_out/r06dp; O=gpurun_out/r06dp; : > $O/summary.txt
for ex in rccl rccl+shard peer3; do
  for rep in 1 2; do
    MVAE_DP_EXCHANGE=${ex%+shard} MVAE_DP_SHARD_OPTIMIZER=$([ "$ex" = "rccl+shard" ] && echo 1 || echo 0) timeout 300 python bench.py --force-dp --no-cpu-baseline --no-extra-configs > $O/${ex}_$rep.json 2> $O/${ex}_$rep.err
    python - $O/${ex}_$rep.json $ex $rep <<'PY' >> gpurun_out/r06dp/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "rep", sys.argv[3], round(d["value"]), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us", d["config"].get("exchange"), d["config"].get("peer_timeouts"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
cat $O/summary.txt
