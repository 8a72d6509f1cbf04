#!/bin/bash
# Fabric traffic and L2 hit rate of the largest contractions of the conv step (one op per process, tools/p3_one.py):
#   tools/pmc_conv_traffic.sh TAG   ->  gpurun_out/prof_TAG/conv_traffic.json   (summarised into profiles/ by tools/summarise_profiles.py)
TAG=${1:-r04}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG/conv_traffic; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for op in ${OPS:-e2f d2f e1f d1f pd2 pd1 pe2 pe1}; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES"; do
    d=$OUT/${op}_$(echo $grp | tr ' ' '_' | cut -c1-24)
    env $EXTRA_ENV timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o g -- python $ROOT/tools/p3_one.py $op > $d.log 2>&1
  done
done
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections, os
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/*/")):
    op = os.path.basename(d.rstrip("/")).split("_")[0]
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc, dur, names = collections.defaultdict(list), [], set()
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"]
        if "k_gemm_p3" in k or "k_gemm_f32pp" in k or "k_gemm_tiled" in k or "k_gemm_b3" in k:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            names.add(k.replace("void ", "").split("(")[0])
            if row["Counter_Name"] in ("FETCH_SIZE", "GRBM_GUI_ACTIVE"):
                dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    r = res.setdefault(op, {"kernel": sorted(names)})
    for c, v in acc.items():
        r[c] = sum(v) / len(v)
    if dur:
        r.setdefault("duration_us_under_pmc", round(sum(dur) / len(dur) / 1e3, 2))
json.dump(res, open(os.path.dirname(out) + "/conv_traffic.json", "w"), indent=1)
for op, r in res.items():
    f, w = r.get("FETCH_SIZE", 0) * 2 * 1024, r.get("WRITE_SIZE", 0) * 1024
    hit = r.get("TCC_HIT_sum", 0) / max(r.get("TCC_HIT_sum", 0) + r.get("TCC_MISS_sum", 0), 1)
    print(f"{op:6s} fetch {f / 1e6:8.1f} MB  write {w / 1e6:8.1f} MB  L2 hit {hit:.2f}  {r.get('kernel')}")
PY
