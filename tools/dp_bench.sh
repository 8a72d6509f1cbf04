# the data-parallel routes at world size 1 (--force-dp): RCCL all-reduce in two overlapped buckets / in one piece, and
# the peer-read exchange (copy + flag kernel + reduction fused into the optimizer launch)
for cfg in "allreduce 1" "allreduce 0" "peer 1"; do
  set -- $cfg
  MVAE_DP_EXCHANGE=$1 MVAE_DP_OVERLAP=$2 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --force-dp > gpurun_out/dp_$1_$2.json 2> gpurun_out/dp_$1_$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/dp_$1_$2.json").read().strip().splitlines()[-1])
print("exchange=$1 overlap=$2", d["value"], d["ms_per_step"], d["config"]["graph_replays"], d["config"]["graph_steps"])
PY
done
