#!/bin/bash
for v in noepi nomma nopro; do
  MVAE_HIP_LIB=$PWD/mvae_amd/_variants/libmvae_hip_$v.so timeout 120 python tools/bench_decode_bce.py 6 2>&1 | tail -1 | sed "s/^/$v /"
done
timeout 120 python tools/bench_decode_bce.py 6 2>&1 | tail -1 | sed "s/^/base /"
