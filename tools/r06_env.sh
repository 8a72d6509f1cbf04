#!/bin/bash
# Runtime-knob sweep for the headline step: the same two bench commands under different HIP / ROCr environment settings.
# Output: gpurun_out/r06env/summary.txt
mkdir -p gpurun_out/r06env; O=gpurun_out/r06env
run() {  # name, env...
  name=$1; shift
  for mode in "20 5" "2000 200"; do
    set -- "$@"; s=${mode% *}; w=${mode#* }
    for rep in 1 2; do
      env "$@" timeout 300 python bench.py --steps $s --warmup $w --no-cpu-baseline --no-extra-configs > $O/${name}_${s}_$rep.json 2> $O/${name}_${s}_$rep.err
      python - "$O/${name}_${s}_$rep.json" "$name" "$s" "$rep" <<'PY' >> gpurun_out/r06env/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "steps", sys.argv[3], "rep", sys.argv[4], round(d["value"]), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us")
except Exception as e:
    print(sys.argv[2], sys.argv[3], sys.argv[4], "FAILED", e)
PY
    done
  done
}
run base X=1
run activewait ROC_ACTIVE_WAIT_TIMEOUT=100000
run nointerrupt HSA_ENABLE_INTERRUPT=0
run nopktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run nosysscope ROC_SYSTEM_SCOPE_SIGNAL=0
run base2 X=1
cat $O/summary.txt
