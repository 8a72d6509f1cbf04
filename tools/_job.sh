export TMPDIR=/tmp
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "edge or step" 2>&1 | tail -2
rm -rf /tmp/p_b; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_b -o r --output-format csv -- python tools/bench_conv.py 256 30 > /tmp/p_b.log 2>&1
f=$(find /tmp/p_b -name '*kernel_stats.csv' | head -1)
python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_edge3' in r['Name']: print(r['Name'][:30], r['Calls'], r['AverageNs'])
PY
GRAPH=1 python tools/bench_conv.py 256 100
