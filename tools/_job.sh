mkdir -p gpurun_out/j17
run() { python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2))
except Exception as e: print('$1', 'ERR', e)"; }
(
run base
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run pktcap0
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run pktcap1
HIP_FORCE_DEV_KERNARG=0 run devkernarg0
HIP_FORCE_DEV_KERNARG=1 run devkernarg1
AMD_OPT_FLUSH=0 run optflush0
AMD_OPT_FLUSH=1 run optflush1
DEBUG_HIP_GRAPH_BATCH_SIZE=1000 run gbatch1000
DEBUG_HIP_GRAPH_BATCH_SIZE=1 run gbatch1
DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 run hdpwa0
AMD_DIRECT_DISPATCH=0 run direct0
GPU_FLUSH_ON_EXECUTION=1 run flushexec1
run base2
) > gpurun_out/j17/env.txt 2>&1
