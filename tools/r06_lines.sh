#!/bin/bash
# the bench lines of the final build, run AFTER profiles/r06_pmc_traffic.json exists so that they quote the counter traffic
out=gpurun_out/r06_lines
mkdir -p $out
timeout 900 python bench.py > $out/r06_bench_line.json 2> $out/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 > $out/r06_bench_driver.json 2> $out/bench_driver.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model 6h2,6s2,6e2 > $out/r06_bench_prod36.json 2>$out/p.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model e6 --fixed-curvature > $out/r06_bench_e6.json 2>$out/e.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --force-dp > $out/r06_bench_forced_dp.json 2>$out/f.err
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model h40 --steps 500 --warmup 50 > $out/r06_bench_h40.json 2>$out/h.err
for f in $out/r06_*.json; do tail -n 1 $f > $f.tmp && mv $f.tmp $f; done
if [ -z "$R06_NO_PYTEST" ]; then timeout 1500 python -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; fi
tail -n 3 $out/pytest_gpu.log
python - <<PY
import json
for f in ("bench_line","bench_driver","bench_prod36","bench_e6","bench_forced_dp","bench_h40"):
    d=json.load(open("$out/r06_%s.json" % f)); r=d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"]*1e3,2), "traffic_step", r.get("traffic_step"), r.get("traffic_source"))
PY
