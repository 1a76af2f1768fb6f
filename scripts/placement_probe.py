"""Placement lottery probe: config 2 runs 14.0-14.5 ms for most placements of its output arrays and 18-20 ms for some.
One process: arrays are allocated afresh several times (spacers perturb where they land); every placement is timed as it is and with the
V / t views shifted inside over-allocated buffers by a few candidate offsets -- does the RELATIVE position of the X and V streams matter?
Usage (GPU box): python scripts/placement_probe.py [reps] > gpurun_out/placement.json"""
import json, sys
sys.path.insert(0, ".")
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
C, d, n_sk = 100000, 50, 1000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x = torch.randn((C, d), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (C, d), device=dev, generator=g).to(torch.float64) * 2 - 1
s = P.ZigZagAD(d, P.GaussianIso())
PAD = 1 << 22                                     # doubles of slack in front of / behind each array (32 MiB)
SHIFTS = [0, 32, 512, 8192, 1 << 18, 1 << 21]     # doubles: 256 B, 4 KiB, 64 KiB, 2 MiB, 16 MiB


def best_of(out, n=4):
    ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ensemble.sample_skeleton_device(s, n_sk, x, v, seed=i, out=out, check=False)
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])


rows = []
spacers = []
for rep in range(reps):
    nX = C * n_sk * d
    bX = torch.empty((nX + 2 * PAD,), dtype=torch.float64, device=dev)
    bV = torch.empty((nX + 2 * PAD,), dtype=torch.float64, device=dev)
    bt = torch.empty((C * n_sk + 2 * PAD,), dtype=torch.float64, device=dev)
    row = {"rep": rep, "X": hex(bX.data_ptr()), "V": hex(bV.data_ptr()), "t": hex(bt.data_ptr()), "ms": {}}
    for sh in SHIFTS:
        out = {"X": bX[PAD:PAD + nX].view(C, n_sk, d), "V": bV[PAD + sh:PAD + sh + nX].view(C, n_sk, d), "t": bt[PAD:PAD + C * n_sk].view(C, n_sk)}
        row["ms"][f"V+{sh * 8}"] = round(best_of(out), 2)
    out = {"X": bX[PAD + 512:PAD + 512 + nX].view(C, n_sk, d), "V": bV[PAD + 512:PAD + 512 + nX].view(C, n_sk, d), "t": bt[PAD:PAD + C * n_sk].view(C, n_sk)}
    row["ms"]["both+4096"] = round(best_of(out), 2)
    out = {"X": bV[PAD:PAD + nX].view(C, n_sk, d), "V": bX[PAD:PAD + nX].view(C, n_sk, d), "t": bt[PAD:PAD + C * n_sk].view(C, n_sk)}
    row["ms"]["swapped"] = round(best_of(out), 2)
    rows.append(row)
    print(json.dumps(row), flush=True)
    del bX, bV, bt, out
    torch.cuda.empty_cache()
    if rep % 2 == 0:
        spacers.append(torch.empty(((rep + 1) * 3 << 26,), dtype=torch.uint8, device=dev))      # hold something: the next arrays land elsewhere
    elif spacers:
        spacers.pop()
