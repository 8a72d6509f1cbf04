#!/bin/bash
# Re-record what depends on the HOST side of the conv step (launch list, step times, bench lines) into gpurun_out/prof_<tag>/
# without repeating the per-contraction PMC passes of tools/collect_profiles.sh (valid while the kernels' source_hash is unchanged).
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/conv $OUT/pmc_mfma_conv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conv -o conv -- python $ROOT/tools/bench_conv.py 256 20 > $OUT/conv.log 2>&1
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $MFMA --kernel-trace --output-format csv -d $OUT/pmc_mfma_conv -o conv -- python $ROOT/tools/bench_conv.py 256 5 > $OUT/pmc_mfma_conv.log 2>&1
cd $ROOT
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 600 python bench.py --no-cpu-baseline --config conv > $OUT/bench_conv.json 2>&1
timeout 600 python bench.py --no-cpu-baseline --config conv --gpus 1 --steps 20 --warmup 5 > $OUT/bench_conv_driver.json 2>&1
(cd /tmp && rm -rf $OUT/epoch && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/epoch -o epoch -- python $ROOT/tools/bench_epoch.py > $OUT/epoch.log 2>&1)   # launch list of an epoch: no prepare launch per step
timeout 300 python tools/bench_epoch.py > $OUT/bench_epoch.txt 2>&1   # a 60000-image MNIST epoch through the device-side input pipeline
MVAE_FEED_FOLD=0 timeout 300 python tools/bench_epoch.py > $OUT/bench_epoch_pairs.txt 2>&1   # the same as [prepare, step] pairs
find $OUT -name '*kernel_trace.csv' -size +20M -delete
tail -c 300 $OUT/bench_conv.json
