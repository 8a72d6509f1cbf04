#!/bin/bash
# long-run stability: the CLI (train_stopping: up to 100 epochs) on synthetic MNIST, several seeds, batch 100 (padded rows) and 128
cd $GRAFT_REPO_ROOT
for seed in 2 3 4 5 6 7 8 9; do
for cfg in "b100 100" "b128 128"; do
  set -- $cfg
  r=$( ( timeout 900 python -m mvae_amd.run --model h2,s2,e2 --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size $2 --seed $seed ) 2>&1 | grep -v amdgpu.ids | grep 'TrainEpoch\|non-finite' | tail -1 | cut -c1-100 )
  echo "seed $seed $1: $r"
done
done
python tools/eps_zero_scan.py 9 128 20 2 2>&1 | tail -1
python tools/eps_zero_scan.py 8 100 90 2>&1 | tail -1
