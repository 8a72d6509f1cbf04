export TMPDIR=/tmp
for v in base nomask base nomask; do
  if [ $v = base ]; then unset MVAE_HIP_LIB; else export MVAE_HIP_LIB=$PWD/mvae_amd/_variants/libmvae_hip_$v.so; fi
  rm -rf /tmp/p_$v; MVAE_BENCH_ALLOW_NONFINITE=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$v -o r --output-format csv -- python tools/bench_conv.py 256 30 > /tmp/p_$v.log 2>&1
  f=$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_gemm_p3' in r['Name']: print(r['Name'][:75], r['Calls'], r['AverageNs'])
PY
done
