#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01b
# Writes everything under gpurun_out/prof_<tag>/; tools/summarise_profiles.py turns that into profiles/<tag>_*.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-configs"
PMCB="python $ROOT/bench.py --steps 300 --warmup 50 --graph-steps 0 --no-cpu-baseline --no-extra-configs"
cd /tmp
# 1. kernel trace + stats (graph replays, as bench.py runs by default)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o bench -- $BENCH > $OUT/ktrace.log 2>&1
# 2./3. PMC passes, one counter each, kernel trace only
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $PMCB > $OUT/pmc_write.log 2>&1
# 3b. the same two passes for the other MLP kernel paths (block kernels: configs[3]; wave-cooperative components: h40; e6 =
#     configs[0]), so that their bench legs can quote counter traffic of THEIR kernels
for cfg in "prod36 --model 6h2,6s2,6e2" "h40 --model h40" "e6 --model e6 --fixed-curvature"; do
  set -- $cfg; key=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${key}_$ctr -o bench -- $PMCB "$@" > $OUT/pmc_${key}_$ctr.log 2>&1
  done
done
# 4. conv architecture (BASELINE configs[4]) kernel stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conv -o conv -- python $ROOT/tools/bench_conv.py 256 20 > $OUT/conv.log 2>&1
# 4b. MFMA-pipe busy cycles (own PMC pass, kernel trace only): the MLP step and the conv step
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $MFMA --kernel-trace --output-format csv -d $OUT/pmc_mfma -o bench -- $PMCB > $OUT/pmc_mfma.log 2>&1
timeout 600 rocprofv3 --pmc $MFMA --kernel-trace --output-format csv -d $OUT/pmc_mfma_conv -o conv -- python $ROOT/tools/bench_conv.py 256 5 > $OUT/pmc_mfma_conv.log 2>&1
cd $ROOT
# 4c. the conv contractions one by one: fabric traffic, L2 hit rate, MFMA-busy (-> conv_traffic.json)
bash $ROOT/tools/pmc_conv_traffic.sh $TAG > $OUT/conv_traffic.log 2>&1
cd $ROOT
# 5. the un-profiled bench lines of the same build
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err   # the default invocation: configs [1] + the legs of [0], [3], [4] + CPU rows
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err   # as the driver calls it
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model 6h2,6s2,6e2 > $OUT/bench_prod36.json 2>&1   # learnable curvature, as golden mnist_prod36_learn
timeout 600 python bench.py --no-cpu-baseline --config conv > $OUT/bench_conv.json 2>&1
timeout 600 python bench.py --no-cpu-baseline --config conv --gpus 1 --steps 20 --warmup 5 > $OUT/bench_conv_driver.json 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model e6 --fixed-curvature > $OUT/bench_e6.json 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --force-dp > $OUT/bench_forced_dp.json 2>&1   # world 1, exchange forced (librccl)
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model h40 --steps 500 --warmup 50 > $OUT/bench_h40.json 2>&1
# 6. the log-likelihood estimator (scope row f-1): kernel stats of one call sequence, the counters of its fused decoder launch
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loglik -o ll -- python $ROOT/tools/bench_ll.py > $OUT/loglik.log 2>&1)
bash $ROOT/tools/pmc_decode_bce.sh ${TAG}_loglik > $OUT/loglik_pmc.log 2>&1
cp $ROOT/gpurun_out/prof_${TAG}_loglik/pmc.txt $OUT/loglik_pmc.txt 2>/dev/null
timeout 120 python tools/bench_decode_bce.py > $OUT/loglik_decoder.txt 2>&1
# keep only the small summaries (the traces are large)
(cd /tmp && rm -rf $OUT/epoch && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/epoch -o epoch -- python $ROOT/tools/bench_epoch.py > $OUT/epoch.log 2>&1)   # launch list of an epoch: no prepare launch per step
timeout 300 python tools/bench_epoch.py > $OUT/bench_epoch.txt 2>&1   # a 60000-image MNIST epoch through the device-side input pipeline
MVAE_FEED_FOLD=0 timeout 300 python tools/bench_epoch.py > $OUT/bench_epoch_pairs.txt 2>&1   # the same as [prepare, step] pairs
find $OUT -name '*kernel_trace.csv' -size +20M -delete
ls -la $OUT $OUT/*/ 2>/dev/null | head -60
