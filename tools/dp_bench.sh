for ov in 1 0; do
  MVAE_DP_OVERLAP=$ov python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --force-dp > gpurun_out/dp_ov$ov.json 2> gpurun_out/dp_ov$ov.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/dp_ov$ov.json").read().strip().splitlines()[-1])
print("overlap=$ov", d["value"], d["ms_per_step"], d["config"]["graph_replays"], d["config"]["graph_steps"])
PY
done
