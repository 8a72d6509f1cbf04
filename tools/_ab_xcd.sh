for v in main xcd; do
  if [ "$v" = main ]; then unset MVAE_HIP_LIB; else export MVAE_HIP_LIB=$PWD/mvae_amd/_variants/libmvae_hip_$v.so; fi
  echo "== $v"; python tools/bench_gemm.py 2>/dev/null | grep -E "^(e1|e2|d1|d2)"; python tools/bench_split.py 2>/dev/null | grep -E "mode 1"
  for m in 0 1; do MVAE_CONV_SPLIT_BF16=$m python tools/bench_conv.py 256 60 2>/dev/null | tail -1; done
done
