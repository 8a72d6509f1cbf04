#!/bin/bash
# rocprofv3 kernel stats of the conv step (BASELINE configs[4], B=256): tools/prof_conv.sh <tag>
TAG=${1:-conv}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conv -o conv -- python $ROOT/tools/bench_conv.py 256 25 > $OUT/conv.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/conv/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
steps = 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'].split('(')[0][:70]:72s} calls/step {int(r['Calls'])/steps:5.1f} us/step {float(r['TotalDurationNs'])/steps/1e3:8.1f}")
print("total us/step", tot / steps / 1e3)
PY
