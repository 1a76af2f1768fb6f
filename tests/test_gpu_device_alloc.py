"""pdmp_device_alloc / ensemble.alloc_chunked: output arrays built from independent physical chunks (csrc/device_alloc.hip) -- plain
device memory as far as the engine and torch are concerned: same results as arrays from torch's allocator, ragged sizes, release."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu


def test_chunked_outputs_equal_plain_outputs(engine, oracle):
    import torch
    from pdmpflux.jl_amd import ensemble
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    Cn, n_sk = 300, 97
    x, v = make_init("ZigZagAD", 50, Cn, seed=3)
    dev = torch.device("cuda:0")
    xd, vd = torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev)
    plain = ensemble.sample_skeleton_device(s, n_sk, xd, vd, seed=5)
    # tiny chunks (2 MiB) so that the arrays span several of them and the last one is ragged
    out = {"X": ensemble.alloc_chunked((Cn, n_sk, 50), chunk_bytes=2 << 20), "V": ensemble.alloc_chunked((Cn, n_sk, 50), chunk_bytes=4 << 20),
           "t": ensemble.alloc_chunked((Cn, n_sk), chunk_bytes=2 << 20)}
    assert out["X"].is_cuda and out["X"].is_contiguous() and out["X"].dtype == torch.float64
    got = ensemble.sample_skeleton_device(s, n_sk, xd, vd, seed=5, out=out)
    torch.cuda.synchronize()
    for k in ("X", "V", "t"):
        assert torch.equal(got[k], plain[k])
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 5, diagnostics=False)
    assert_bitwise({k: got[k].cpu().numpy() for k in ("X", "V", "t")}, ref, keys=("X", "V", "t"))
    # torch ops on the chunked memory (reductions, slicing, copies) behave like on any tensor
    assert torch.equal(out["t"][:, -1].clone().cpu(), plain["t"][:, -1].cpu())
    del out, got
    torch.cuda.synchronize()


def test_device_alloc_abi(engine):
    import torch
    from pdmpflux.jl_amd import _lib, ensemble
    L = _lib.lib()
    torch.cuda.init()
    free0 = torch.cuda.mem_get_info()[0]
    p = C.c_void_p()
    assert L.pdmp_device_alloc(-1, 3 * (1 << 30) + 12345, 0, C.byref(p)) == 0 and p.value
    assert torch.cuda.mem_get_info()[0] <= free0 - 3 * (1 << 30)
    assert L.pdmp_device_free(p) == 0
    assert L.pdmp_device_free(p) == _lib.ERR_ARGUMENT                    # not (any more) a pointer of pdmp_device_alloc
    assert b"not a pointer" in L.pdmp_device_alloc_error()
    assert L.pdmp_device_alloc(-1, 0, 0, C.byref(p)) == _lib.ERR_ARGUMENT
    assert L.pdmp_device_free(None) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)            # the chunks went back to the driver
    # default policy of the host API: arrays of at least 1 GiB are chunked, smaller ones come from torch
    o = ensemble.alloc_outputs(3000, 1000, 50)
    assert o["X"].numel() * 8 >= ensemble.CHUNKED_FROM_BYTES and o["t"].numel() * 8 < ensemble.CHUNKED_FROM_BYTES
    o["X"].fill_(1.5); o["t"].zero_()
    assert float(o["X"].sum().item()) == 1.5 * o["X"].numel()
