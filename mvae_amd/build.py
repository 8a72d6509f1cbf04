"""Builds libmvae_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

The library is eight translation units (csrc/mvae_api.hip, mvae_step.hip, mvae_conv.hip, mvae_p3.hip, mvae_f32pp.hip, mvae_edge.hip, mvae_peer.hip, mvae_rccl.hip) compiled in parallel into
csrc/_obj/*.o and linked; a unit is recompiled only when it or a header is newer than its object.

    python -m mvae_amd.build [--force] [--timing]      (--timing: the -DMV_DBG_TIMING build used by tools/phase_timing.py,
                                                        written to libmvae_hip_timing.so)
"""
import os
from typing import Optional
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["mvae_api", "mvae_step", "mvae_conv", "mvae_p3", "mvae_f32pp", "mvae_edge", "mvae_peer", "mvae_rccl"]
HEADERS = [os.path.join(CSRC, h) for h in ("mvae_common.hpp", "mvae_math.hpp", "mvae_gemm.hpp", "mvae_fastmath.hpp", "mvae_step_blk.hpp", "mvae_coop.hpp", "mvae_p3.hpp")] + \
          [os.path.join(os.path.dirname(HERE), "include", "mvae_hip.h")]
DEPS = [os.path.join(CSRC, u + ".hip") for u in UNITS] + HEADERS
LIB = os.path.join(HERE, "libmvae_hip.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-value",
         # a/b and sqrt lower to v_rcp_f32 / v_sqrt_f32 sequences (<= 2.5 ulp) instead of the ~10-instruction
         # correctly-rounded expansions: the manifold chain is latency-bound and the parity bar is 1e-4
         "-fno-hip-fp32-correctly-rounded-divide-sqrt",
         # subnormal f32 inputs/outputs of VALU ops flush to zero (MFMA C/D never flush): removes the frexp/ldexp
         # range scaling around every v_rcp_f32 / v_exp_f32 / v_log_f32; and a/b may become a * (1/b)
         "-fgpu-flush-denormals-to-zero", "-freciprocal-math"]


def source_hash() -> str:
    """Fingerprint of the HIP sources the library is built from (sha1 over the .hip units and the headers, in a fixed
    order).  profiles/*_pmc_traffic.json record it; bench.py only quotes counter summaries of the build it is running."""
    import hashlib
    h = hashlib.sha1()
    for path in DEPS:
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


CONV_UNITS = ("mvae_conv", "mvae_p3", "mvae_f32pp", "mvae_edge")


def file_hashes(paths=None) -> dict:
    """{file name: sha1 of its bytes} of the sources (default: all of DEPS).  The conv counter summaries record them: a conv
    kernel's machine code depends on its own translation unit and the headers only (conv_source_hash)."""
    import hashlib
    out = {}
    for path in (paths or DEPS):
        with open(path, "rb") as fh:
            out[os.path.basename(path)] = hashlib.sha1(fh.read()).hexdigest()[:16]
    return out


def conv_file_hashes() -> dict:
    """file_hashes of what the conv engine's kernels are compiled from: the four conv translation units and every header."""
    return file_hashes([os.path.join(CSRC, u + ".hip") for u in CONV_UNITS] + HEADERS)


def lib_is_fresh(lib: str = LIB) -> bool:
    return os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in DEPS)


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmvae_hip.so")
    return hipcc


def build(force: bool = False, verbose: bool = False, timing: bool = False) -> str:
    lib = LIB if not timing else os.path.join(HERE, "libmvae_hip_timing.so")
    if not force and lib_is_fresh(lib):
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "_obj_timing" if timing else "_obj")
    os.makedirs(objdir, exist_ok=True)
    extra = ["-DMV_DBG_TIMING"] if timing else []
    newest_header = max(os.path.getmtime(h) for h in HEADERS)

    def compile_unit(u: str) -> str:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header):
            return obj
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp, obj)
        return obj

    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
        objs = list(pool.map(compile_unit, UNITS))
    tmp = f"{lib}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, lib)  # atomic: concurrent builders (one per rank) never expose a partial file
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timing="--timing" in sys.argv))


_LLVM = "/opt/rocm/lib/llvm/bin"
_ISA_CACHE: dict = {}


def _isa_of_elf(elf: str) -> dict:
    """{kernel base name: fingerprint over the machine code of all of its instantiations} of one device ELF: per function
    the disassembly's instruction text + encodings (addresses dropped), per base name the sorted (symbol, hash) pairs."""
    import hashlib
    import re
    import subprocess
    text = subprocess.check_output([os.path.join(_LLVM, "llvm-objdump"), "-d", elf], text=True)
    per, name, buf = {}, None, []

    def close():
        if name:
            per[name] = hashlib.sha256("".join(buf).encode()).hexdigest()[:16]
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            close()
            name, buf = m.group(1), []
        elif name:
            t = line.split("//")
            enc = t[1].split(":", 1)[1].strip() if len(t) > 1 and ":" in t[1] else ""
            buf.append(t[0].strip() + "|" + enc + "\n")
    close()
    groups = {}
    for sym, hsh in per.items():
        m = re.match(r"^_Z(\d+)", sym)
        base = sym[m.end():m.end() + int(m.group(1))] if m else sym
        groups.setdefault(base, []).append(sym + ":" + hsh)
    return {b: hashlib.sha256("\n".join(sorted(v)).encode()).hexdigest()[:16] for b, v in groups.items()}


def kernel_isa(unit: str = "mvae_step", objdir: Optional[str] = None) -> Optional[dict]:
    """Machine-code fingerprints of the kernels of one translation unit, taken from the object the library was linked from
    (csrc/_obj/<unit>.o: its gfx950 code object is extracted and disassembled with the ROCm install's llvm tools).  Counter
    files record them next to `source_hash`: a summary collected from ANOTHER revision of the sources still describes a
    launch whose machine code is identical in this build (bench.py).  None when the object or the tools are missing."""
    import subprocess
    import tempfile
    obj = os.path.join(objdir or os.path.join(CSRC, "_obj"), unit + ".o")
    tools = [os.path.join(_LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(obj) or not all(os.path.exists(t) for t in tools):
        return None
    key = (obj, os.path.getmtime(obj), os.path.getsize(obj))
    if key in _ISA_CACHE:  # (~5 s per unit: once per process)
        return _ISA_CACHE[key]
    _ISA_CACHE[key] = None
    try:
        with tempfile.TemporaryDirectory() as d:
            fat, elf = os.path.join(d, "fat.bin"), os.path.join(d, "dev.elf")
            # (an explicit output file: without one llvm-objcopy rewrites its INPUT in place)
            subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            subprocess.check_call([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + fat, "--output=" + elf], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _ISA_CACHE[key] = _isa_of_elf(elf)
            return _ISA_CACHE[key]
    except (OSError, subprocess.CalledProcessError, ValueError):
        return None
