#!/bin/bash
# full GPU suite + the bench lines of the current build (gpurun_out/r06full)
out=gpurun_out/r06full
mkdir -p $out
timeout 1500 python -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a $out/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --force-dp > $out/bench_forced_dp.json 2> $out/bench_forced_dp.err
python - <<PY | tee -a $out/summary.txt
import json
for f in ("bench_driver", "bench_forced_dp"):
    try:
        d=json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), "steps/s", round(d["ms_per_step"]*1e3,2), "us", d.get("configs_summary"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -n 6 $out/pytest_gpu.log
