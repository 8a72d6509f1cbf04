mkdir -p gpurun_out/j3
V=mvae_amd/_variants
python -m pytest tests/test_hip_parity.py tests/test_model_api_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/j3/pytest.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/j3/early_$i.json 2>/dev/null
MVAE_HIP_LIB=$PWD/$V/libmvae_hip_late.so python bench.py --no-cpu-baseline --no-extra-configs > gpurun_out/j3/late_$i.json 2>/dev/null
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/j3/drv_early.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-prewarm > gpurun_out/j3/drv_early_nopre.json 2>/dev/null
MVAE_HIP_LIB=$PWD/$V/libmvae_hip_early_t.so python tools/phase_timing.py > gpurun_out/j3/phase_early.txt 2>&1
MVAE_HIP_LIB=$PWD/$V/libmvae_hip_late_t.so python tools/phase_timing.py > gpurun_out/j3/phase_late.txt 2>&1
