#!/bin/bash
# end-to-end stability sweep: the CLI for up to 100 epochs on the synthetic stand-in data sets, several models / seeds;
# prints the last epoch line (or the error) of each run
cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  r=$( ( timeout 900 python -m mvae_amd.run "$@" ) 2>&1 | grep -v amdgpu.ids | grep 'TrainEpoch\|Error\|error\|non-finite\|Traceback' | tail -1 | cut -c1-150 )
  echo "$name: $r"
}
for seed in 11 12 13; do
  run "prod36 b100 s$seed" --model 6h2,6s2,6e2 --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size 100 --seed $seed
  run "e6 b100 s$seed" --model e6 --fixed_curvature True --epochs 4 --likelihood_n 0 --batch_size 100 --seed $seed
  run "s2 b100 s$seed" --model s2 --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size 100 --seed $seed
  run "h2 b128 s$seed" --model h2 --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size 128 --seed $seed
  run "d2,p2 b100 s$seed" --model d2,p2 --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size 100 --seed $seed
  run "3u2 universal s$seed" --model 3u2 --universal True --fixed_curvature False --epochs 40 --likelihood_n 0 --batch_size 100 --seed $seed
  run "h2,s2,e2 scalar s$seed" --model h2,s2,e2 --scalar_parametrization True --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size 100 --seed $seed
  run "h2,s2,e2 fixed s$seed" --model h2,s2,e2 --fixed_curvature True --epochs 4 --likelihood_n 0 --batch_size 100 --seed $seed
done
run "conv cifar s1" --dataset cifar --architecture conv --h_dim 8192 --batch_size 256 --model h2,s2,e2 --fixed_curvature False --epochs 4 --likelihood_n 0 --seed 1
run "h40 s1" --model h40 --fixed_curvature False --epochs 4 --likelihood_n 0 --batch_size 128 --seed 1
