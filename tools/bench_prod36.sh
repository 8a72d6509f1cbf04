#!/bin/bash
# config [3] (6h2,6s2,6e2, learnable curvature): parity tests of the block kernels, phase timing, bench line
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "block_latent or full_size" 2>&1 | grep -E "Error|passed|failed" | cut -c1-600 | head -5
MVAE_HIP_LIB=mvae_amd/libmvae_hip_timing.so timeout 300 python tools/phase_timing.py 6h2,6s2,6e2 2>&1 | grep -E "fwd23: (loads|heads|comp|hd)|latent_|statistics"
timeout 300 python bench.py --model 6h2,6s2,6e2 --steps 2000 --warmup 200 --no-cpu-baseline | tee gpurun_out/prod36_learn.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
[ -n "$ROWS" ] && MVAE_BLK_FWD=0 timeout 300 python bench.py --model 6h2,6s2,6e2 --steps 2000 --warmup 200 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('row forward:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
exit 0
