"""The hand-written fp64 MFMA product of the dense-precision forms (csrc/dgemm_mfma.hip: Out = P [x | v], v_mfma_f64_16x16x4_f64) against
torch's fp64 matmul: every tile path (128- and 64-column tiles), ragged M / N / K edges, and the batched dense forms run through it
(PDMP_DENSE_GEMM=mfma) against the same forms through rocBLAS."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_init, make_pair

pytestmark = pytest.mark.gpu


def _gemm(M, N, K):
    import torch
    from pdmpflux.jl_amd import _lib
    L = _lib.lib()
    L.pdmp_dgemm_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    g = torch.Generator(device="cuda"); g.manual_seed(M * 7 + N * 3 + K)
    At = torch.randn((K, M), dtype=torch.float64, device="cuda", generator=g)      # column-major M x K
    Bt = torch.randn((N, K), dtype=torch.float64, device="cuda", generator=g)      # column-major K x N
    Ct = torch.full((N, M), float("nan"), dtype=torch.float64, device="cuda")      # column-major M x N
    rc = L.pdmp_dgemm_mfma(M, N, K, At.data_ptr(), M, Bt.data_ptr(), K, Ct.data_ptr(), M, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    ref = Bt @ At
    err = (Ct - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)
    return err


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (128, 128, 8), (130, 70, 10), (1000, 137, 1000), (998, 64, 500), (256, 4096, 64),
                                   (1000, 2000, 1000), (1000, 20000, 1000), (2, 3, 2), (1024, 1, 1024), (126, 65, 18)])
def test_dgemm_mfma_matches_matmul(M, N, K):
    assert _gemm(M, N, K) < 5e-13


def test_dgemm_mfma_refuses_odd_dimensions():
    """M, K (and the leading dimensions) must be even -- 16-byte loads of row / k pairs; the dense forms send an odd d to rocBLAS."""
    import torch
    from pdmpflux.jl_amd import _lib
    L = _lib.lib()
    L.pdmp_dgemm_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    a = torch.zeros(64, dtype=torch.float64, device="cuda")
    assert L.pdmp_dgemm_mfma(7, 3, 4, a.data_ptr(), 8, a.data_ptr(), 4, a.data_ptr(), 8, None) != 0
    assert L.pdmp_dgemm_mfma(8, 3, 3, a.data_ptr(), 8, a.data_ptr(), 4, a.data_ptr(), 8, None) != 0


def test_dense_forms_through_the_mfma_product(engine, oracle, monkeypatch):
    """Zig-Zag (incremental form), BPS and Boomerang on a dense precision with the product P [x | v] by the hand-written kernel: the same
    trajectories as through rocBLAS to the tolerance the batched forms are held to (the two GEMMs sum k in different orders)."""
    for kind, n_sk in (("ZigZagAD", 60), ("BPS", 25), ("BoomerangAD", 25)):
        outs = {}
        for impl in ("rocblas", "mfma"):
            monkeypatch.setenv("PDMP_DENSE_GEMM", impl)
            s, cfg = make_pair(engine, oracle, kind, 192, "dense")
            x, v = make_init(kind, 192, 80, seed=4)
            outs[impl] = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=12)
        a, b = outs["rocblas"], outs["mfma"]
        assert np.allclose(a["t"], b["t"], rtol=1e-8, atol=1e-10), kind
        assert np.allclose(a["X"], b["X"], rtol=1e-7, atol=1e-9), kind
