#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the MLP step (two PMC passes) -> gpurun_out/prof_$1/pmc_{fetch,write}
TAG=${1:-tmp}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PMCB="python $ROOT/bench.py --steps 300 --warmup 50 --graph-steps 0 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $PMCB > $OUT/pmc_write.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for c, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = glob.glob("gpurun_out/prof_$TAG/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == c and row["Kernel_Name"].startswith(("void k_", "k_")):
            acc[row["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, round(sum(v) / len(v), 1), "KB")
PY
