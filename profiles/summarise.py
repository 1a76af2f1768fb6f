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
out = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_wr", "pmc_rd"):
    for fcsv in newest(f"{src}/{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(fcsv)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                meta = {k: r[k] for k in ("Kernel_Name", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
        out.update(agg)
        out["_dispatch"] = meta
if "WRITE_SIZE" in out:
    out["hbm_write_bytes"] = out["WRITE_SIZE"] * 1024            # KiB; exact for wide streaming stores (MI355X_MICROARCH.md, HBM)
if "FETCH_SIZE" in out:
    out["hbm_read_bytes_corrected"] = out["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE reports half of a coalesced stream
out["_note"] = "one launch of the dominant kernel (bench.py --steps 1 --warmup 0); each --pmc group collected in its own run"
json.dump(out, open(f"{dest}/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
