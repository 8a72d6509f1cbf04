"""Dev tool: A/B builds of the library.  python tools/build_variant.py NAME [-DFLAG=V ...] [--units mvae_step,mvae_conv]
Compiles the named units (default: mvae_step) with the extra defines, links them with the main build's other objects into
mvae_amd/_variants/libmvae_hip_NAME.so (select with MVAE_HIP_LIB=...).  Add --timing for the -DMV_DBG_TIMING build."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import build as b
name = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
units = ["mvae_step"]
for i, a in enumerate(sys.argv):
    if a == "--units":
        units = sys.argv[i + 1].split(",")
if "--timing" in sys.argv:
    defs.append("-DMV_DBG_TIMING")
    units = list(b.UNITS)
b.build()  # main objects up to date
vdir = os.path.join(b.HERE, "_variants")
os.makedirs(vdir, exist_ok=True)
objs = []
procs = []
for u in b.UNITS:
    if u in units:
        obj = os.path.join(vdir, f"{u}_{name}.o")
        procs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + defs + ["-c", os.path.join(b.CSRC, u + ".hip"), "-o", obj],
                                      stderr=subprocess.DEVNULL))
    else:
        obj = os.path.join(b.CSRC, "_obj", u + ".o")
    objs.append(obj)
for p in procs:
    assert p.wait() == 0
lib = os.path.join(vdir, f"libmvae_hip_{name}.so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
