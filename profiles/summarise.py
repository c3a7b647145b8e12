"""Summarise a profiles/collect.sh run: per-kernel duration table (from --stats) and HBM
traffic per launch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE is in KB and counts half of the bytes
of wide coalesced loads)."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]


def find(sub, pat):
    f = glob.glob(f"{out}/{sub}/**/*{pat}", recursive=True)
    return f[0] if f else None


def short(name):
    n = name.replace("void ", "").replace("mgs::", "")
    return n.split("(")[0]


stats = find("stats", "kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(f"{out}/kernel_stats.csv", "w") as fh:
    fh.write("kernel,calls,avg_us,total_ms,percent\n")
    for r in rows:
        fh.write(f"{short(r['Name'])},{r['Calls']},{float(r['AverageNs']) / 1e3:.2f},"
                 f"{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['Percentage']):.2f}\n")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
traffic, raw = {}, {}
with open(f"{out}/pmc_fetch_write_per_kernel.txt", "w") as fh:
    fh.write("kernel launches FETCH_SIZE_KB WRITE_SIZE_KB traffic_MB=(2*FETCH+WRITE)*1024 uncorrected_MB\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
        if k.startswith("at::") or "rocclr" in k:
            continue
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        traffic[k] = int((2 * fe + wr) * 1024)
        raw[k] = int((fe + wr) * 1024)
        fh.write(f"{k} {len(v['FETCH_SIZE'])} {fe:.1f} {wr:.1f} {(2 * fe + wr) * 1024 / 1e6:.1f} {(fe + wr) * 1024 / 1e6:.1f}\n")
def bench_name(k):
    """kernel symbol (templates stripped by short()) -> the name bench.py's kernels_us uses"""
    if k.startswith("k_blend_bwd<false, false, false>") or k == "k_blend_bwd<false>":
        return "blend_bwd"
    if k.startswith("k_blend_fwd"):
        return "blend_fwd"
    if k.startswith("k_preprocess_bwd"):
        return "preprocess_bwd"
    if k.startswith("k_tile_sort<1024"):
        return "tile_sort"
    if k.startswith("k_bin_lds"):
        return "bin_lds"          # count (with the projection) and emit launches averaged
    if k.startswith("k_bin_colsum"):
        return "bin_colsum"
    return None


names = {k: bench_name(k) for k in traffic if bench_name(k)}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_tree_hash  # noqa: E402  (the same hash bench.py checks before quoting these bytes)
json.dump({"workload": "SYN-C 300000 @ 640x480",
           "csrc_tree_hash": csrc_tree_hash(),
           "source": "profiles/collect.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
           "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes: gfx950 FETCH_SIZE counts half of wide "
                         "coalesced loads (MI355X_MICROARCH.md, HBM)",
           "bytes_per_launch": {names[k]: v for k, v in traffic.items() if k in names},
           "uncorrected_bytes_per_launch": {names[k]: raw[k] for k in traffic if k in names}},
          open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(open(f"{out}/kernel_stats.csv").read())
print(open(f"{out}/pmc_fetch_write_per_kernel.txt").read())


# optional: SQ counter passes (profiles/collect.sh step 4) -> per-kernel averages per launch
if len(sys.argv) > 2 and sys.argv[2] == "sq":
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/sq_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{out}/sq_counters_per_kernel.txt", "w") as fh:
        for k, v in sorted(sq.items()):
            if not (k.startswith("k_blend") or k.startswith("k_preprocess_bwd") or k.startswith("k_tile_sort<1024")):
                continue
            fh.write(k + "\n")
            for c, vals in sorted(v.items()):
                fh.write(f"  {c:28s} {sum(vals) / len(vals):16.0f}  (avg of {len(vals)} launches)\n")
    print(open(f"{out}/sq_counters_per_kernel.txt").read())
    # VALU figures of the timed kernels into pmc_traffic.json (bench.py quotes roofline.valu_busy from there, under the
    # same source hash as the traffic).  valu_busy = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * kernel cycles): the counter adds
    # up, over all waves, the (quad-)cycles a wave spent issuing VALU instructions - a lone wave issues one per ~4.4
    # cycles, a SIMD with 6-8 resident waves retires one per 1.2-2.3 cycles (profiles/r02_valu_issue_microbench.txt),
    # so 1.0 means "one wave's worth of back-to-back VALU issue per SIMD all the time", not "the pipe is full".
    # Kernel cycles = rocprofv3 --stats duration * 2.4 GHz (nominal clock), 1024 SIMDs.
    dur = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    tj = json.load(open(f"{out}/pmc_traffic.json"))
    tj["valu"] = {}
    for k, v in sq.items():
        b = bench_name(k)
        if b is None or k not in dur or "SQ_ACTIVE_INST_VALU" not in v:
            continue
        act = sum(v["SQ_ACTIVE_INST_VALU"]) / len(v["SQ_ACTIVE_INST_VALU"])
        insts = sum(v["SQ_INSTS_VALU"]) / len(v["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in v else None
        cycles = dur[k] * 2400.0
        tj["valu"][b] = {"SQ_ACTIVE_INST_VALU": int(act), "SQ_INSTS_VALU": None if insts is None else int(insts),
                         "avg_us": round(dur[k], 2), "valu_busy": round(act * 4 / (1024 * cycles), 3),
                         "cycles_per_valu_per_simd": None if not insts else round(1024 * cycles / insts, 2)}
    tj["valu_note"] = ("valu_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * duration * 2.4 GHz): wave-issue cycles summed over "
                       "the waves per SIMD cycle (a lone wave: one VALU per ~4.4 cycles; a full SIMD retires one per 1.2-2.3)")
    json.dump(tj, open(f"{out}/pmc_traffic.json", "w"), indent=1)
