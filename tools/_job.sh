mkdir -p gpurun_out/j13
python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/j13/pytest.txt
for m in h40 "h40,s40,e40"; do
python bench.py --model $m --no-cpu-baseline --no-extra-configs --steps 500 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$m', round(d['value']), round(d['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})"
done > gpurun_out/j13/large.txt 2>&1
