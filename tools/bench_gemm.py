"""Dev tool: TFLOP/s of the dense contractions on the conv architecture's layer shapes (B = 256)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import functional as Fn
from mvae_amd.conv import _gemm_nn, _gemm_tn
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


shapes = [("e0", 65536, 64, 48), ("e1", 16384, 128, 1024), ("e2", 4096, 512, 2048), ("d1", 4096, 4096, 128),
          ("d2", 16384, 1024, 256), ("d3", 65536, 48, 64)]
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # batch multiplier (B = 256 * scale)
for name, M, N, K in shapes:
    M *= scale
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    Wn = torch.randn(K, N, device=dev)
    Q = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    a = timeit(lambda: Fn.linear_forward(x, W, None))
    b = timeit(lambda: _gemm_nn(x, Wn))
    c = timeit(lambda: _gemm_tn(x, Q))
    print(f"{name} M={M} N={N} K={K}:  NT {a*1e6:7.1f} us {fl/a/1e12:6.1f} TF | NN {b*1e6:7.1f} us {fl/b/1e12:6.1f} TF | "
          f"TN {c*1e6:7.1f} us {fl/c/1e12:6.1f} TF")
