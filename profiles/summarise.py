#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (profiles/collect.sh) into profiles/<round>/<tag>_{kernel_stats.csv,pmc.json}."""
import collections, csv, glob, json, os, shutil, sys

tag, dest = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "pdmp::"          # which kernels count (substring of the kernel name)
src = f"gpurun_out/prof_{tag}"
os.makedirs(dest, exist_ok=True)
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]      # a tag collected twice: the last run counts
stats = newest(f"{src}/trace/*/*_kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(f"{dest}/{tag}_kernel_stats.csv", "w", newline="") as f:
        csv.writer(f).writerows([rows[0]] + [r for r in rows[1:] if pat in r[0]])
if os.path.exists(f"{src}/bench_traced.json"):
    shutil.copy(f"{src}/bench_traced.json", f"{dest}/{tag}_bench_traced.json")
# The DOMINANT kernel of the launch = the matching kernel with the largest total duration in the kernel trace (a launch of the headline
# workload also enqueues lane_check_kernel and a group kernel that returns at once): its dispatches alone are summed, `_dispatch`
# describes it, and whatever the other matching kernels counted is listed per kernel under `_other_kernels` -- round 3's version summed
# every `pdmp::` kernel and described whichever came last in the file.
def short(name):
    return name.split("(")[0].strip()
dominant = None
if stats:
    best = None
    for r in rows[1:]:
        if pat in r[0]:
            try:
                tot = float(r[rows[0].index("TotalDurationNs")]) if "TotalDurationNs" in rows[0] else float(r[2])
            except (ValueError, IndexError):
                continue
            if best is None or tot > best[0]:
                best = (tot, short(r[0]))
    dominant = best[1] if best else None
out, others = {}, {}
for d in ("pmc_sq", "pmc_sq2", "pmc_wr", "pmc_rd"):
    for fcsv in newest(f"{src}/{d}/*/*_counter_collection.csv"):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        meta = {}
        for r in csv.DictReader(open(fcsv)):
            if pat in r["Kernel_Name"]:
                k = short(r["Kernel_Name"])
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
                meta[k] = {q: r[q] for q in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if q in r}
        if not per:
            continue
        dom = dominant if dominant in per else max(per, key=lambda k: per[k].get("GRBM_GUI_ACTIVE", 0.0) or sum(per[k].values()))
        out.update(per[dom])
        out["_dispatch"] = meta[dom]
        for k in per:
            if k != dom:
                others.setdefault(k, {}).update(per[k])
if others:
    out["_other_kernels"] = others
if "WRITE_SIZE" in out:
    out["hbm_write_bytes"] = out["WRITE_SIZE"] * 1024            # KiB; exact for wide streaming stores (MI355X_MICROARCH.md, HBM)
if "FETCH_SIZE" in out:
    out["hbm_read_bytes_corrected"] = out["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE reports half of a coalesced stream
out["_note"] = "the dispatches of the dominant kernel (`_dispatch`) of ONE launch (bench.py --steps 1 --warmup 0); each --pmc group collected in its own run; other kernels of the launch under `_other_kernels`"
json.dump(out, open(f"{dest}/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
