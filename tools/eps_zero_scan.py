"""Dev check: does the device input pipeline ever draw an eps pair that is exactly (0, 0) for a component (Box-Muller with
u1 = 1)?  Scans the batches of a seeded run: python tools/r06_l.py SEED BATCH EPOCHS"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from mvae_amd.runner import EpochRunner
seed, B, epochs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # cursor values an epoch consumes, in units of nb (2 when a ragged tail batch exists)
dev = torch.device("cuda:0")
eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
images = (torch.rand(60000, 784, device=dev) * 255).to(torch.uint8)
os.environ["MVAE_NO_PAD_ROWS"] = "1"
er = EpochRunner(eng, images, B, seed=seed, fold=False)
hits = 0
for cur in [e * stride * er.nb + b for e in range(epochs) for b in range(er.nb)]:
    eng.counters[8] = cur
    er._prepare()
    e = er._bufs[0][1][:B]
    z = (e.view(B, 3, 2) == 0).all(dim=2)
    if bool(z.any()):
        rows = z.nonzero().tolist()
        hits += len(rows)
        print(f"cursor {cur} (epoch {cur // (stride * er.nb)}, batch {cur % er.nb}): exact-zero eps pair at (row, component) {rows}")
print(f"seed {seed} batch {B}: {hits} exact-zero pairs in {epochs * er.nb} batches")
