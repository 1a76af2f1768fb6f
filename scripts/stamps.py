import sys, os, shutil, ctypes as C
sys.path.insert(0, '.')
# diagnostic build: swap the library in a scratch copy of the package dir is not needed -- load it by path
import pdmpflux.jl_amd._lib as L
L.LIB_PATH = os.path.abspath("pdmpflux.jl_amd/libpdmphip_stamp.so")   # make -C pdmpflux.jl_amd/csrc stamp (git-ignored; travels with gpurun)
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble
dev = torch.device("cuda:0")
C_, d, n_sk = int(sys.argv[2]) if len(sys.argv) > 2 else 100000, 50, 1000
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((C_, d), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (C_, d), device=dev, generator=g).to(torch.float64) * 2 - 1
kind = sys.argv[1] if len(sys.argv) > 1 else "zz"
if kind == "bps":
    x = 1.0 + 0.1 * x
    v = torch.randn((C_, d), dtype=torch.float64, device=dev, generator=g); v = (v / v.norm(dim=1, keepdim=True)).contiguous()
    s = P.BPS(d, P.Banana())
elif kind == "boom":
    C_, d = (int(sys.argv[2]) if len(sys.argv) > 2 else 50000), 100
    x = torch.randn((C_, d), dtype=torch.float64, device=dev, generator=g)
    v = torch.randn((C_, d), dtype=torch.float64, device=dev, generator=g)
    s = P.Boomerang(d, P.GaussianIso())
else:
    s = P.ZigZagAD(d, P.GaussianIso())
out = ensemble.sample_skeleton_device(s, n_sk, x, v, seed=1)
torch.cuda.synchronize()
lib = L.lib()
buf = (C.c_ulonglong * 8)()
lib.pdmp_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
print("rc", lib.pdmp_debug_stamps(buf))
vals = list(buf)[:6]; print("cells", buf[6], "wave-bounds", buf[7], "cells per wave-bound", buf[6]/max(buf[7],1))
tot = sum(vals)

names = ["BOUND(exp+cells+inversion)", "true_rate+ar", "uniform+flow_state", "velocity_jump", "record", "loop top/other"]
if s.last_launch_form() == 4:       # lane kernel (csrc/lane_kernel.hpp, LANE_MARK)
    names = ["prefetch + E draw", "bound cells + epilogue", "rate + accept + walk", "flow + flip", "record: row to LDS", "record: tiles (reads, stores)",
             "record: carry, times, diag", "loop tail"]
    vals = list(buf)[:8]; tot = sum(vals)
for n, v_ in zip(names, vals):
    print(f"{n:32s} {v_:>16d}  {100*v_/tot:5.1f}%")
