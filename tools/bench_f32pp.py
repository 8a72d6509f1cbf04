"""Dev tool: the exact-f32 NT contraction on large square-ish shapes, per tile variant of k_gemm_f32pp (MVAE_F32PP_TILE) and on
the register-staged k_gemm_tiled (mvae_set_forward_kernel(0)): steady-state TFLOP/s without the small-shape effects."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mvae_amd import functional as Fn
    from mvae_amd._lib import load
    load().mvae_set_contraction_mode(0)
    load().mvae_set_forward_kernel(int(sys.argv[2]))
    dev = torch.device("cuda:0")
    for M, N, K in ((8192, 4096, 4096), (16384, 512, 2048), (4096, 512, 2048), (16384, 128, 1024)):
        x, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        for _ in range(3):
            Fn.linear_forward(x, W, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            Fn.linear_forward(x, W, None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"   M={M} N={N} K={K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
else:
    for label, env, k in (("k_gemm_tiled", {}, 0), ("f32pp 128x128x32", {"MVAE_F32PP_TILE": "0"}, 1),
                          ("f32pp 128x64x32", {"MVAE_F32PP_TILE": "1"}, 1), ("f32pp 128x64x64", {"MVAE_F32PP_TILE": "2"}, 1)):
        print(label)
        subprocess.run([sys.executable, __file__, "child", str(k)], env=dict(os.environ, **env))
