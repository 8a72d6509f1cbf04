#!/bin/bash
# Dev tool: A/B of library variants (tools/build_variant.py) on the contraction shapes and the conv step.
# usage: tools/ab_gemm.sh name1 name2 ...   ("main" = the in-tree build)
for v in "$@"; do
  if [ "$v" = main ]; then unset MVAE_HIP_LIB; else export MVAE_HIP_LIB=$PWD/mvae_amd/_variants/libmvae_hip_$v.so; fi
  echo "== $v"
  python tools/bench_gemm.py 2>/dev/null | grep -E "^(e1|e2|d1|d2)"
  python tools/bench_conv.py 256 60 2>/dev/null | tail -1
done
