#!/bin/bash
# PMC passes over one backward contraction of the conv step (tools/p3_one.py OP [b3]): tools/pmc_p3.sh TAG OP [b3] -> gpurun_out/prof_TAG/pmc.txt
TAG=${1:-p3}; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/pmc.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  d=$OUT/g_$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o g -- python $ROOT/tools/p3_one.py "$@" > $d.log 2>&1
  python - "$d" >> $OUT/pmc.txt <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no output for", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(fs[0])):
    if "k_gemm_p3" in row["Kernel_Name"] or "k_gemm_b3" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:40s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cd $ROOT; cat $OUT/pmc.txt
