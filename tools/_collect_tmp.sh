timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/full_suite_r05.txt
bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1
python tools/summarise_profiles.py r05 > gpurun_out/summarise_r05.log 2>&1
mkdir -p gpurun_out/r05_profiles
cp gpurun_out/prof_r05/loglik/*kernel_stats.csv profiles/r05_loglik_kernel_stats.csv
cp gpurun_out/prof_r05/loglik_pmc.txt profiles/r05_loglik_decoder_pmc.txt
cp gpurun_out/prof_r05/loglik_decoder.txt profiles/r05_loglik_decoder.txt
# the MLP bench lines once more, now that the counter summaries of THIS build exist
timeout 900 python bench.py > profiles/r05_bench_line.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > profiles/r05_bench_driver.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model 6h2,6s2,6e2 > profiles/r05_bench_prod36.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --model e6 --fixed-curvature > profiles/r05_bench_e6.json 2>/dev/null
cp profiles/r05_* gpurun_out/r05_profiles/
rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_loglik
cat gpurun_out/full_suite_r05.txt; tail -3 gpurun_out/summarise_r05.log
