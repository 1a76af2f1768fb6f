"""fp64 GEMM P (d x d) x B (d x N): the hand-written MFMA kernel (csrc/dgemm_mfma.hip) against rocBLAS (through torch.matmul), d = 1000."""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import torch
from pdmpflux.jl_amd import _lib
L = _lib.lib()
L.pdmp_dgemm_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
d = 1000
for N in (10000, 20000, 5000, 3334, 2500, 1250):
    At = torch.randn((d, d), dtype=torch.float64, device="cuda")
    Bt = torch.randn((N, d), dtype=torch.float64, device="cuda")
    Ct = torch.empty((N, d), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def mine():
        L.pdmp_dgemm_mfma(d, N, d, At.data_ptr(), d, Bt.data_ptr(), d, Ct.data_ptr(), d, st)
    def blas():
        torch.matmul(Bt, At, out=Ct)
    res = {}
    for name, f in (("mfma_kernel", mine), ("rocblas", blas), ("mfma_kernel_again", mine)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[name] = {"ms": round(ms, 4), "tflops": round(2.0 * d * d * N / ms / 1e9, 2)}
    print(json.dumps({"d": d, "N": N, **res}), flush=True)
