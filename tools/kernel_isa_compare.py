"""Dev tool (CPU only): which kernels of a translation unit have IDENTICAL machine code in two revisions of the sources.

    python tools/kernel_isa_compare.py <old-git-rev> [<new-git-rev> | WORKTREE] [--unit mvae_step]

Both revisions of mvae_amd/csrc/<unit>.hip are compiled device-only for gfx950 with the flags of mvae_amd/build.py (the
headers come from the same revision), disassembled with llvm-objdump, and compared kernel by kernel on instruction text +
encoding (addresses dropped).  Used in round 6 to show that the counter files of profiles/r06_* -- collected one commit before
the last kernel change -- describe the same machine code for every launch of the BASELINE configs (profiles/r06_isa_compare.txt).
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-value",
         "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero", "-freciprocal-math"]


def materialise(rev, dst):
    """The csrc directory and the public header of `rev` (or of the working tree) under dst/, in the repo's layout."""
    os.makedirs(os.path.join(dst, "mvae_amd", "csrc"))
    os.makedirs(os.path.join(dst, "include"))
    if rev == "WORKTREE":
        for f in os.listdir(os.path.join(ROOT, "mvae_amd", "csrc")):
            if f.endswith((".hip", ".hpp")):
                open(os.path.join(dst, "mvae_amd", "csrc", f), "wb").write(open(os.path.join(ROOT, "mvae_amd", "csrc", f), "rb").read())
        open(os.path.join(dst, "include", "mvae_hip.h"), "wb").write(open(os.path.join(ROOT, "include", "mvae_hip.h"), "rb").read())
        return
    names = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "mvae_amd/csrc/"], text=True).split()
    for n in names + ["include/mvae_hip.h"]:
        if n.endswith((".hip", ".hpp", ".h")):
            open(os.path.join(dst, n), "wb").write(subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{n}"]))


def kernel_hashes(rev, unit):
    with tempfile.TemporaryDirectory() as d:
        materialise(rev, d)
        c = os.path.join(d, "mvae_amd", "csrc")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--offload-device-only", "-c", unit + ".hip", "-o", "u.co"], cwd=c,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=u.co", "--output=u.elf"], cwd=c)
        text = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "u.elf"], cwd=c, text=True)
    out, name, buf = {}, None, []

    def close():
        if name:
            out[name] = hashlib.sha256("".join(buf).encode()).hexdigest()[:16]
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
    return out


def demangled_head(sym):
    try:
        return subprocess.check_output([LLVM + "/llvm-cxxfilt", sym], text=True).strip().split("(")[0]
    except Exception:  # noqa: BLE001
        return sym


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    unit = sys.argv[sys.argv.index("--unit") + 1] if "--unit" in sys.argv else "mvae_step"
    if "--unit" in sys.argv:
        args = [a for a in args if a != unit]
    old, new = args[0], (args[1] if len(args) > 1 else "WORKTREE")
    a, b = kernel_hashes(old, unit), kernel_hashes(new, unit)
    same = sorted(k for k in a if k in b and a[k] == b[k])
    diff = sorted(k for k in a if k in b and a[k] != b[k])
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    print(f"# {unit}.hip: {old} ({len(a)} kernels) vs {new} ({len(b)} kernels): identical machine code {len(same)}, "
          f"different {len(diff)}, only in {old}: {len(only_a)}, only in {new}: {len(only_b)}")
    for title, ks in (("IDENTICAL", same), ("DIFFERENT", diff), (f"ONLY IN {old}", only_a), (f"ONLY IN {new}", only_b)):
        print(f"## {title}")
        for k in ks:
            print(f"{a.get(k, b.get(k))}  {demangled_head(k)}")
