#!/bin/bash
# where the time of the plane kernels goes: the same launch with pieces switched off (-DMV_P3_DBG variant library)
export MVAE_HIP_LIB=$(pwd)/mvae_amd/_variants/libmvae_hip_p3dbg.so
for op in ${OPS:-db1 da1}; do
  for dbg in ${DBGS:-0 8 72 104 120 106 108 122 124 14}; do
    MV_P3_DBG=$dbg python - $op <<'PY'
import os, sys, torch
sys.argv = [sys.argv[0], sys.argv[1]]
sys.path.insert(0, os.getcwd())
exec(open("tools/p3_one.py").read().split("for _ in range(20):")[0])
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn(); Cv._DEFERRED_WS.clear()
e1.record(); torch.cuda.synchronize()
d = int(os.environ["MV_P3_DBG"])
names = ["nowait", "noA", "noB", "noMFMA", "hot", "nobarrier", "noreads"]
print(op, "dbg", d, "+".join(n for i, n in enumerate(names) if d >> i & 1) or "full", "us %.1f" % (e0.elapsed_time(e1) / 20 * 1e3))
PY
  done
done
