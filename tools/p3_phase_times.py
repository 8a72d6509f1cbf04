"""Dev tool (needs the -DMV_P3_DBG variant library): cycle stamps of the phases of one K step of k_gemm_p3 in waves 0 (group 0)
and 4 (group 1) of workgroup 0.   MVAE_HIP_LIB=mvae_amd/_variants/libmvae_hip_p3dbg.so python tools/p3_phase_times.py OP"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.getcwd())
op = sys.argv[1] if len(sys.argv) > 1 else "db1"
sys.argv = [sys.argv[0], op]
exec(open("tools/p3_one.py").read().split("for _ in range(20):")[0])
lib = C.CDLL(os.environ["MVAE_HIP_LIB"])
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
assert lib.mvae_p3_debug_stamps(buf) == 0
names = ["L start", "reads issued", "DMA issued", "vm wait (g1)", "lgkm wait", "barrier 1", "MFMAs issued", "vm wait (g0)", "barrier 2"]
for g in range(2):
    t = [buf[g * 16 + i] for i in range(9)]
    print(f"{op} group {g}: " + "  ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 9)) + f"   | step {t[8] - t[0]} cycles")
t0, t1 = [buf[i] for i in range(9)], [buf[16 + i] for i in range(9)]
print(f"   group 1 starts its L {t1[0] - t0[0]} cycles after group 0; (100 MHz clock? compare with the step's wall time)")
