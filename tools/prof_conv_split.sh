#!/bin/bash
# rocprofv3 kernel stats + MFMA-busy PMC pass of the conv step (B = 256) with split-product contractions on:
#   tools/prof_conv_split.sh <tag>   -> gpurun_out/prof_<tag>_split/{conv, pmc}
TAG=${1:-conv}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_${TAG}_split; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export MVAE_CONV_SPLIT_BF16=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conv -o conv -- python $ROOT/tools/bench_conv.py 256 20 > $OUT/conv.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o conv -- python $ROOT/tools/bench_conv.py 256 5 > $OUT/pmc.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, json, collections
f = sorted(glob.glob("$OUT/conv/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
steps = 25
print("total us/step", sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
fc = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(fc)):
    if "k_gemm" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    mfma = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    # busy cycles summed over 1024 SIMDs against the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    out[k] = {"dispatches": len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), "mfma_busy_cycles": mfma, "gui_active_cycles_x8": gui,
              "mfma_util": mfma / 1024.0 / (gui / 8.0)}
    print(f"{k[:60]:62s} MfmaUtil {out[k]['mfma_util']:.3f}")
json.dump(out, open("$OUT/pmc_mfma_split.json", "w"), indent=1)
PY
