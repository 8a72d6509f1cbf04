"""Dev tool (needs the -DMV_F32PP_DBG variant library: python tools/build_variant.py f32dbg -DMV_F32PP_DBG --units mvae_f32pp):
cycle stamps of the phases of one K step of k_gemm_f32pp in waves 0 (group 0) and 4 (group 1) of workgroup 0.
   MVAE_HIP_LIB=mvae_amd/_variants/libmvae_hip_f32dbg.so python tools/f32pp_phase_times.py [e2f|e1f|d2f|d1f|big]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.getcwd())
op = sys.argv[1] if len(sys.argv) > 1 else "e2f"
if op == "big":
    from mvae_amd import functional as Fn
    from mvae_amd._lib import load
    load().mvae_set_contraction_mode(0)
    x, W = torch.randn(8192, 4096, device="cuda"), torch.randn(4096, 4096, device="cuda")
    fn = lambda: Fn.linear_forward(x, W, None)  # noqa: E731
else:
    sys.argv = [sys.argv[0], op]
    exec(open("tools/p3_one.py").read().split("for _ in range(20):")[0])
lib = C.CDLL(os.environ["MVAE_HIP_LIB"])
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
assert lib.mvae_f32pp_debug_stamps(buf) == 0
names = ["L start", "reads + DMA issued", "waits", "barrier 1", "MFMAs issued", "vm wait (g0)", "barrier 2"]
for g in range(2):
    t = [buf[g * 16 + i] for i in range(7)]
    print(f"{op} group {g}: " + "  ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 7)) + f"   | step {t[6] - t[0]} cycles")
print("   group 1 starts its L", buf[16] - buf[0], "cycles after group 0")
