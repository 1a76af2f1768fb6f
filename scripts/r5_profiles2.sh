#!/bin/bash
# rocprofv3 passes of config 2 (and the banana) on the round's final lane kernels
bash profiles/collect.sh r5d_zz_iso_lane || exit 1
python3 profiles/summarise.py r5d_zz_iso_lane gpurun_out/r5 || exit 1
bash profiles/collect.sh r5d_zz_banana_lane --target banana || exit 1
python3 profiles/summarise.py r5d_zz_banana_lane gpurun_out/r5 || exit 1
python3 - <<'PY'
import json
for tag in ("r5d_zz_iso_lane", "r5d_zz_banana_lane"):
    d = json.load(open(f"gpurun_out/r5/{tag}_pmc.json"))
    ev = 100000 * 999
    print(tag, {k: round(v / ev, 3) for k, v in d.items() if isinstance(v, float) and k.startswith("SQ_INSTS")}, "written GB %.2f" % (d["hbm_write_bytes"] / 1e9), "read GB %.2f" % (d["hbm_read_bytes_corrected"] / 1e9), "busy %.3f" % (4 * d["SQ_ACTIVE_INST_VALU"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    print(open(f"gpurun_out/r5/{tag}_kernel_stats.csv").read().splitlines()[1][:200])
PY
