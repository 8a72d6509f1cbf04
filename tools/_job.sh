export TMPDIR=/tmp
for m in warm cold; do
rm -rf /tmp/p_$m; MODE=$m timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$m -o r --output-format csv -- python tests/dev/conv_latent_repeat.py > /tmp/p_$m.log 2>&1
f=$(find /tmp/p_$m -name '*kernel_stats.csv' | head -1)
echo == $m
python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_cl_' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
