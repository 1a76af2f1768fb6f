""""a handle is not re-entrant; distinct handles may be used from distinct host threads" -- for the batched dense-precision
forms too (round 5: their rocBLAS handles, side stream and events belong to the engine handle; rounds 1-4 shared them per device, so two
dense BPS / Boomerang handles on one device raced on rocblas_set_stream).  Two host threads, one handle and one stream each, several
launches each, at once -- against the same launches run one after the other."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _precision(d, seed):
    rs = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rs.normal(size=(d, d)))
    P = (Q * np.exp(rs.uniform(np.log(0.1), np.log(10.0), size=d))) @ Q.T
    return 0.5 * (P + P.T)


@pytest.mark.parametrize("kind,gemm", [("BPS", "rocblas"), ("BPS", "mfma"), ("BoomerangAD", "rocblas"), ("ZigZagAD", "mfma")])
def test_two_threads_two_dense_handles(engine, monkeypatch, kind, gemm):
    import torch
    from pdmpflux.jl_amd import ensemble
    monkeypatch.setenv("PDMP_DENSE_GEMM", gemm)
    P = engine
    d, C, n_sk, reps = 64, 256, 40, 4
    dev = torch.device("cuda:0")
    rs = np.random.default_rng(3)
    x = torch.from_numpy(rs.normal(size=(C, d))).to(dev)
    v = rs.normal(size=(C, d))
    if kind == "ZigZagAD":
        v = np.sign(v)
    elif kind == "BPS":
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = torch.from_numpy(v).to(dev)
    kw = {"refresh_rate": 0.5} if kind != "ZigZagAD" else {}
    samplers = [getattr(P, kind)(d, P.GaussianPrecision(_precision(d, 11 + q)), **kw) for q in range(2)]

    def run(q, stream, out):
        for r in range(reps):
            o = ensemble.sample_skeleton_device(samplers[q], n_sk, x, v, seed=100 * q + r, stream=stream, check=True)
            out.append({k: o[k].clone() for k in ("X", "V", "t")})

    seq = [[], []]
    for q in range(2):                                                    # the reference: one handle after the other, default stream
        run(q, None, seq[q])
    torch.cuda.synchronize()
    assert samplers[0].last_launch_form() == 5                            # the batched dense form
    par = [[], []]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    errs = []

    def worker(q):
        try:
            with torch.cuda.stream(streams[q]):
                run(q, streams[q], par[q])
            streams[q].synchronize()
        except Exception as e:                                             # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(q,)) for q in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for q in range(2):
        for r in range(reps):
            for k in ("X", "V", "t"):
                assert torch.equal(par[q][r][k], seq[q][r][k]), (kind, gemm, q, r, k)
