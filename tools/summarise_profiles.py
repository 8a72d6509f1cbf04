"""Turn gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the small tracked
summaries under profiles/:  python tools/summarise_profiles.py r01b"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def first(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


def short(name):  # "void k_latent_bwd<2, true, true>(CompTable, ...)" -> "k_latent_bwd"
    name = name.replace("void ", "")
    for stop in "<(":
        if stop in name:
            name = name[:name.index(stop)]
    return name.strip()


def pmc_average(path, counter):
    """rocprofv3 counter_collection.csv: one row per (dispatch, counter) -> {kernel: (mean value, dispatches)}"""
    acc, n = defaultdict(float), defaultdict(int)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k] += float(row["Counter_Value"])
            n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


for sub, out in (("ktrace/**/*kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"),
                 ("ktrace/**/*domain_stats.csv", f"{tag}_bench_domain_stats.csv"),
                 ("conv/**/*kernel_stats.csv", f"{tag}_conv_b256_kernel_stats.csv")):
    f = first(sub)
    if f:
        shutil.copy(f, os.path.join(dst, out))
        print("copied", out)
for name in ("bench_line.json", "bench_prod36.json", "bench_e6.json"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, f"{tag}_{name}"), "w") as fh:
                fh.write(lines[-1] + "\n")
            print("copied", name)
convlog = os.path.join(src, "conv.log")
if os.path.exists(convlog):
    lines = [l for l in open(convlog).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{tag}_conv_b256_line.json"), "w").write(lines[-1] + "\n")

fetch, write = first("pmc_fetch/**/*counter_collection.csv"), first("pmc_write/**/*counter_collection.csv")
if fetch and write:
    F, W = pmc_average(fetch, "FETCH_SIZE"), pmc_average(write, "WRITE_SIZE")
    kernels = {}
    for k in F:
        if not k.startswith("k_"):
            continue
        fkb, n = F[k]
        wkb = W.get(k, (0.0, 0))[0]
        kernels[k] = {"FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1), "dispatches": n,
                      "traffic_bytes": int(round((2.0 * fkb + wkb) * 1024))}
    rec = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace on `python bench.py "
                  "--steps 300 --warmup 50 --graph-steps 0 --no-cpu-baseline`, MI355X, per-dispatch averages",
        "units": "FETCH_SIZE / WRITE_SIZE are KB per dispatch as rocprofv3 reports them",
        "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts 64 B per 128 B request, i.e. "
                      "reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE taken as is "
                      "(uncalibrated per the guide). Both count L2<->fabric traffic, Infinity-Cache hits included, so "
                      "this is an upper bound on HBM bytes (the whole working set is ~12 MB and MALL-resident).",
        "kernels": kernels}
    with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(kernels, indent=1))
