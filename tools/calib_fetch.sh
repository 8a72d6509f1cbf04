#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/calib_fetch.hip) -> gpurun_out/calib_fetch.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/calib; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o calib -- $ROOT/tools/_bin/calib_fetch > $OUT/$c.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/calib/{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, "KB per dispatch:", [round(x, 1) for x in v])
PY
