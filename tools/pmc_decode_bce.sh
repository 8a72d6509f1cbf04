#!/bin/bash
# PMC passes over the fused log-likelihood decoder (tools/bench_decode_bce.py): tools/pmc_decode_bce.sh TAG -> gpurun_out/prof_TAG/pmc.txt
TAG=${1:-gemm}; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/pmc.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INST_LEVEL_LDS"; do
  d=$OUT/g_$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o g -- python $ROOT/tools/bench_decode_bce.py > $d.log 2>&1
  python - "$d" >> $OUT/pmc.txt <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no output for", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(fs[0])):
    if "decode_bce" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:40s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cd $ROOT; cat $OUT/pmc.txt
