"""Dev tool: one contraction shape, repeated (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import functional as Fn
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 512, 2048)
dev = torch.device("cuda:0")
from mvae_amd._lib import load
load().mvae_set_contraction_mode(1 if os.environ.get("MVAE_CONV_SPLIT_BF16") == "1" else 0)
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
for _ in range(20):
    Fn.linear_forward(x, W, None)
torch.cuda.synchronize()
