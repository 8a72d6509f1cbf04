#!/bin/bash
# final profile set of the round: collect on the box, reduce on the box (the raw rocprofv3 output is > 64 MiB), bring back the summaries
TAG=${1:-r06}
bash tools/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
MVAE_DBR32=1 bash tools/pmc_decode_bce.sh ${TAG}_dbr32 > gpurun_out/pmc_dbr32.log 2>&1
python tools/summarise_profiles.py $TAG > gpurun_out/summarise_$TAG.log 2>&1
mkdir -p gpurun_out/${TAG}_profiles
cp profiles/${TAG}_* gpurun_out/${TAG}_profiles/ 2>/dev/null
cp gpurun_out/prof_${TAG}_dbr32/pmc.txt gpurun_out/${TAG}_profiles/${TAG}_loglik_decoder32_pmc.txt 2>/dev/null
cp gpurun_out/prof_$TAG/loglik_pmc.txt gpurun_out/${TAG}_profiles/${TAG}_loglik_decoder_pmc.txt 2>/dev/null
cp gpurun_out/prof_$TAG/loglik_decoder.txt gpurun_out/${TAG}_profiles/${TAG}_loglik_decoder.txt 2>/dev/null
# keep the small logs / lines, drop the raw traces
mkdir -p gpurun_out/${TAG}_logs
find gpurun_out/prof_$TAG -maxdepth 1 -type f -size -2M -exec cp {} gpurun_out/${TAG}_logs/ \;
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_dbr32 gpurun_out/prof_${TAG}_loglik
du -sh gpurun_out; ls gpurun_out/${TAG}_profiles | head -40; tail -n 5 gpurun_out/summarise_$TAG.log
