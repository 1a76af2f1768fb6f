#!/bin/bash
# rocprofv3 passes of config 2 on the 128-byte-line lane kernel
bash profiles/collect.sh r5b_zz_iso_lane || exit 1
python3 profiles/summarise.py r5b_zz_iso_lane gpurun_out/r5 && python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5/r5b_zz_iso_lane_pmc.json"))
ev = 100000 * 999
print({k: round(v / ev, 3) for k, v in d.items() if isinstance(v, float) and k.startswith("SQ_INSTS")}, "written GB %.2f" % (d["hbm_write_bytes"] / 1e9), "busy %.3f" % (4 * d["SQ_ACTIVE_INST_VALU"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
