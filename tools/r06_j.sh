#!/bin/bash
out=gpurun_out/r06j
mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_input_pipeline_gpu.py -x -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/summary.txt
tail -n 6 $out/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $out/summary.txt
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from mvae_amd.runner import EpochRunner
dev = torch.device("cuda:0")
comps = [("h", 2)] * 6 + [("s", 2)] * 6 + [("e", 2)] * 6
images = (torch.rand(60000, 784, device=dev) ** 3 * 255).to(torch.uint8)
for pad in ("1", "0"):
    os.environ["MVAE_NO_PAD_ROWS"] = "0" if pad == "1" else "1"
    eng = StepEngine(comps, 784, 400, dev, radius_trainable=[l != "e" for l, _ in comps])
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
    er = EpochRunner(eng, images, 100, seed=1)
    for _ in range(2): n = er.run_epoch(1.0, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): n = er.run_epoch(1.0, True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"6h2,6s2,6e2 batch 100, buffers {er.Bp} rows ({eng.kernel_path(er.Bp)}): {dt / 5 / n * 1e6:.1f} us/step, elbo/sample {eng.read_stats()['last']['elbo'] / 100:.2f}")
PY
