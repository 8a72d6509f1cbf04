"""Dev tool: steps/s of a whole training epoch through the device-side input pipeline (EpochRunner: gather + dynamic
binarisation + eps draw + fused step per batch, replayed as HIP graphs) on a synthetic 60000 x 784 uint8 set."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from mvae_amd.runner import EpochRunner
dev = torch.device("cuda:0")
comps = [("h", 2), ("s", 2), ("e", 2)]
eng = StepEngine(comps, 784, 400, dev, radius_trainable=[True, True, False], lr=1e-3)
eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
images = (torch.rand(60000, 784, device=dev) ** 3 * 255).to(torch.uint8)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128  # (100 = the reference CLI's default: padded to 112 rows, MVAE_NO_PAD_ROWS=1: exact)
er = EpochRunner(eng, images, BATCH, seed=1)
for _ in range(2):
    n = er.run_epoch(1.0, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
E = 10
for _ in range(E):
    n = er.run_epoch(1.0, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"batch {BATCH} (buffers of {er.Bp} rows, kernels: {eng.kernel_path(er.Bp)}): " if BATCH != 128 else "", end="")
print(f"epoch of {n} steps: {dt / E * 1e3:.2f} ms = {n * E / dt:.0f} steps/s ({dt / E / n * 1e6:.1f} us/step)")
