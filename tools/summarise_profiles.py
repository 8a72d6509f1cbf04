"""Turn gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the small tracked
summaries under profiles/:  python tools/summarise_profiles.py r01b"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvae_amd.build import conv_file_hashes, kernel_isa, source_hash  # noqa: E402
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
                 ("conv/**/*kernel_stats.csv", f"{tag}_conv_b256_kernel_stats.csv"),
                 ("epoch/**/*kernel_stats.csv", f"{tag}_epoch_kernel_stats.csv")):
    f = first(sub)
    if f:
        shutil.copy(f, os.path.join(dst, out))
        print("copied", out)
for name in ("bench_line.json", "bench_driver.json", "bench_prod36.json", "bench_e6.json", "bench_conv.json", "bench_conv_driver.json",
             "bench_forced_dp.json", "bench_h40.json"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, f"{tag}_{name}"), "w") as fh:
                fh.write(lines[-1] + "\n")
            print("copied", name)
for name in ("bench_epoch.txt", "bench_epoch_pairs.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        lines = [l for l in open(f).read().splitlines() if l.startswith("epoch of")]
        if lines:
            open(os.path.join(dst, f"{tag}_{name}"), "w").write(lines[-1] + "\n")
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
        "source_hash": source_hash(),  # bench.py refuses this summary once the kernels' sources change ...
        "kernel_isa": {"unit": "mvae_step", "hashes": kernel_isa("mvae_step")},  # ... unless the launches' machine code did not
        "kernels": kernels}
    # the other kernel paths (tools/collect_profiles.sh step 3b): per-config kernel tables under "configs"
    rec["configs"] = {}
    for key in ("prod36", "h40", "e6"):
        f2, w2 = first(f"pmc_{key}_FETCH_SIZE/**/*counter_collection.csv"), first(f"pmc_{key}_WRITE_SIZE/**/*counter_collection.csv")
        if not (f2 and w2):
            continue
        F2, W2 = pmc_average(f2, "FETCH_SIZE"), pmc_average(w2, "WRITE_SIZE")
        rec["configs"][key] = {k: {"FETCH_SIZE_KB": round(F2[k][0], 1), "WRITE_SIZE_KB": round(W2.get(k, (0.0, 0))[0], 1),
                                   "dispatches": F2[k][1],
                                   "traffic_bytes": int(round((2.0 * F2[k][0] + W2.get(k, (0.0, 0))[0]) * 1024))}
                               for k in F2 if k.startswith("k_")}
    with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(kernels, indent=1))


def mfma_table(path):
    """Per kernel: MFMA-pipe busy cycles (summed over the 1024 SIMDs), dispatch duration, utilisation."""
    acc, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].replace("void ", "")
            name = name[:name.index("(")] if "(" in name else name
            if not name.startswith("k_"):
                continue
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[name] += 1
                dur[name] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    out = {}
    for k in acc:
        m = max(n[k], 1)
        busy, d_ns = acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / m, dur[k] / m
        out[k] = {"dispatches": n[k], "duration_us": round(d_ns / 1e3, 2), "mfma_busy_cycles": int(busy),
                  "f32_mfma_instructions": int(busy / 32), "waves": int(acc[k]["SQ_WAVES"] / m),
                  "mfma_util": round(busy / (d_ns * 2.4 * 1024), 4) if d_ns > 0 else None}
    return out


mf, mfc = first("pmc_mfma/**/*counter_collection.csv"), first("pmc_mfma_conv/**/*counter_collection.csv")
if mf or mfc:
    rec = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace "
                     "(own pass) on `bench.py --steps 300 --warmup 50 --graph-steps 0 --no-cpu-baseline` (mlp_step) and "
                     "`tools/bench_conv.py 256 5` (conv_step); per-dispatch averages",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration_ns * 2.4 GHz * 1024 SIMDs).  Calibration: the "
                      "counter advances 32 cycles per v_mfma_f32_16x16x4_f32 (k_enc_fwd: 200 WG x 8 waves x 28 MFMAs x "
                      "32 = 1 433 600, exactly the measured value), i.e. 100 % = the 157.3 TFLOP/s f32 peak; "
                      "GRBM_GUI_ACTIVE / 8 XCDs over the dispatch duration gives 2.4-2.5 GHz on the long kernels.  "
                      "Durations are those of the serialised PMC run (longer than in the graph replays).",
           "mlp_step": mfma_table(mf) if mf else None, "conv_step": mfma_table(mfc) if mfc else None}
    with open(os.path.join(dst, f"{tag}_pmc_mfma.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print("wrote", f"{tag}_pmc_mfma.json")

# the conv contractions, one op per process (tools/pmc_conv_traffic.sh): fabric traffic, L2 hit rate, MFMA-pipe busy fraction
ct = os.path.join(src, "conv_traffic.json")
if os.path.exists(ct):
    raw = json.load(open(ct))
    SHAPES = {  # op -> (what, algorithmic bytes at B = 256: operands read once + result written once; planes are 6 B per f32)
        "e2f": ("e2 forward, implicit conv, f32", 256 * 64 * 128 * 4 + 512 * 2048 * 4 + 4096 * 512 * 4),
        "d2f": ("d2 forward, implicit transposed conv, f32", 16384 * 256 * 4 + 256 * 1024 * 4 + 65536 * 64 * 4),
        "e1f": ("e1 forward, implicit conv, f32", 256 * 256 * 64 * 4 + 128 * 1024 * 4 + 16384 * 128 * 4),
        "d1f": ("d1 forward, implicit transposed conv, f32", 4096 * 128 * 4 + 128 * 4096 * 4 + 16384 * 256 * 4),
        "db1": ("backward-data of d2 (implicit conv on planes)", 65536 * 64 * 6 + 256 * 1024 * 6 + 16384 * 256 * 10),
        "da1": ("backward-data of e2 (product on planes)", 4096 * 512 * 6 + 512 * 2048 * 6 + 4096 * 2048 * 4),
        "dt0": ("backward-data of d1 (implicit conv on planes, 8 K slices)", 16384 * 256 * 6 + 128 * 4096 * 6 + 8 * 4096 * 128 * 4),
        "da0": ("backward-data of e1 (implicit transposed conv on planes)", 16384 * 128 * 6 + 128 * 1024 * 6 + 65536 * 64 * 4),
        "dWd1": ("weight gradient of d1 (planes, split over rows)", 4096 * 128 * 6 + 16384 * 256 * 6 + 8 * 128 * 4096 * 4),
        "dWe1": ("weight gradient of e1 (planes, split over rows)", 16384 * 128 * 6 + 65536 * 64 * 6 + 32 * 128 * 1024 * 4),
        "dWe2": ("weight gradient of e2 (planes, 4 row slices)", 4096 * 512 * 6 + 256 * 64 * 128 * 6 + 4 * 512 * 2048 * 4),
        "dWd2": ("weight gradient of d2 (planes, 16 row slices)", 16384 * 256 * 6 + 65536 * 64 * 6 + 16 * 256 * 1024 * 4),
        # the step's paired launches: operands as planes (6 B per value) read once, results written once (planes 6 B, f32 4 B,
        # weight-gradient slices as the launch writes them)
        "pd2": ("d2: weight gradient (8 row slices) + backward-data (planes + column sums), ONE launch",
                16384 * 256 * 6 + 2 * 65536 * 64 * 6 + 256 * 1024 * 6 + 8 * 256 * 1024 * 4 + 16384 * 256 * 6),
        "pd1": ("d1: weight gradient + backward-data in 4 K slices, ONE launch",
                4096 * 128 * 6 + 2 * 16384 * 256 * 6 + 128 * 4096 * 6 + 4 * 128 * 4096 * 4 + 4 * 4096 * 128 * 4),
        "pe2": ("e2: weight gradient + implicit backward-data (planes + column sums), ONE launch",
                2 * 4096 * 512 * 6 + 16384 * 128 * 6 + 512 * 2048 * 6 + 2 * 512 * 2048 * 4 + 16384 * 128 * 6),
        "pe1": ("e1: weight gradient + backward-data (f32 + column sums), ONE launch",
                2 * 16384 * 128 * 6 + 65536 * 64 * 6 + 128 * 1024 * 6 + 16 * 128 * 1024 * 4 + 65536 * 64 * 4),
    }
    kern = {}
    for op, r in raw.items():
        f, w = r.get("FETCH_SIZE"), r.get("WRITE_SIZE")
        hit, miss = r.get("TCC_HIT_sum"), r.get("TCC_MISS_sum")
        busy, act = r.get("SQ_VALU_MFMA_BUSY_CYCLES"), r.get("GRBM_GUI_ACTIVE")
        e = {"kernel": r.get("kernel"), "what": SHAPES.get(op, ("", None))[0], "algorithmic_bytes": SHAPES.get(op, ("", None))[1],
             "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "duration_us_under_pmc": r.get("duration_us_under_pmc")}
        if f is not None and w is not None:
            e["traffic_bytes"] = int(round((2.0 * f + w) * 1024))
            if e["algorithmic_bytes"]:
                e["traffic_over_algorithmic"] = round(e["traffic_bytes"] / e["algorithmic_bytes"], 2)
        if hit is not None and miss is not None and hit + miss > 0:
            e["l2_hit_rate"] = round(hit / (hit + miss), 3)
        if busy is not None and act:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy cycles over the 1024 SIMDs
            e["mfma_busy_frac"] = round(busy / (act / 8.0 * 1024), 4)
        kern[op] = e
    rec = {"source": "tools/pmc_conv_traffic.sh: rocprofv3 --pmc <one group> --kernel-trace, one contraction per process "
                     "(tools/p3_one.py, B = 256, shapes of the conv step), groups FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS/"
                     "EA0_RDREQ/REQ | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES in separate passes; "
                     "per-dispatch averages of the contraction kernel only",
           "correction": "FETCH_SIZE doubled (gfx950 counts 64 B per 128 B request, MI355X_MICROARCH.md section HBM); WRITE_SIZE "
                         "as reported; both are L2<->fabric bytes, Infinity-Cache hits included (upper bound on HBM bytes)",
           "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the fraction of SIMD cycles "
                             "with the matrix pipe busy (f32 MFMA 16x16x4: 32 cycles, bf16 16x16x32: 16 cycles per instruction)",
           "source_hash": source_hash(), "conv_file_hashes": conv_file_hashes(), "kernels": kern}
    with open(os.path.join(dst, f"{tag}_conv_pmc_traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print("wrote", f"{tag}_conv_pmc_traffic.json")
