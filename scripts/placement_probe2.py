"""Placement lottery, second probe: is a slow placement slow for ANY kernel (a plain fill of the same arrays), and does carving X, V, t out
of ONE allocation behave differently from three allocations?  Usage (GPU box): python scripts/placement_probe2.py [reps]"""
import json, sys
sys.path.insert(0, ".")
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
C, d, n_sk = 100000, 50, 1000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn((C, d), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (C, d), device=dev, generator=g).to(torch.float64) * 2 - 1
s = P.ZigZagAD(d, P.GaussianIso())
# the same bytes as 10^6 chains x 100 events: a wave's 64 chains then lie within 2.6 MB of each array instead of 25.6 MB
C2, n2 = 10 * C, n_sk // 10
x2 = torch.randn((C2, d), dtype=torch.float64, device=dev, generator=g)
v2 = torch.randint(0, 2, (C2, d), device=dev, generator=g).to(torch.float64) * 2 - 1


def ev_time(fn, n=3):
    ts = []
    for i in range(n + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return round(min(ts[1:]), 2)


nX, nt = C * n_sk * d, C * n_sk
spacers = []
for rep in range(reps):
    one = rep % 2 == 1
    if one:
        big = torch.empty((2 * nX + nt + 3 * 512,), dtype=torch.float64, device=dev)
        bX, bV, bt = big[:nX], big[nX + 512:2 * nX + 512], big[2 * nX + 1024:2 * nX + 1024 + nt]
    else:
        bX = torch.empty((nX,), dtype=torch.float64, device=dev); bV = torch.empty((nX,), dtype=torch.float64, device=dev)
        bt = torch.empty((nt,), dtype=torch.float64, device=dev)
    out = {"X": bX.view(C, n_sk, d), "V": bV.view(C, n_sk, d), "t": bt.view(C, n_sk)}
    row = {"rep": rep, "one_allocation": one, "X": hex(bX.data_ptr()),
           "kernel_ms": ev_time(lambda i: ensemble.sample_skeleton_device(s, n_sk, x, v, seed=i, out=out, check=False)),
           "short_rows_ms": ev_time(lambda i: ensemble.sample_skeleton_device(s, n2, x2, v2, seed=i, out={"X": bX.view(C2, n2, d), "V": bV.view(C2, n2, d), "t": bt.view(C2, n2)}, check=False)),
           "fill_X_ms": ev_time(lambda i: bX.fill_(float(i))), "fill_V_ms": ev_time(lambda i: bV.fill_(float(i))),
           "copy_X_to_V_ms": ev_time(lambda i: bV.copy_(bX)),
           "kernel_again_ms": ev_time(lambda i: ensemble.sample_skeleton_device(s, n_sk, x, v, seed=i, out=out, check=False))}
    print(json.dumps(row), flush=True)
    del bX, bV, bt, out
    if one:
        del big
    torch.cuda.empty_cache()
    if rep % 3 == 0:
        spacers.append(torch.empty(((rep + 1) * 5 << 26,), dtype=torch.uint8, device=dev))
    elif spacers and rep % 3 == 2:
        spacers.pop()
