"""PDMPHistory (src/Composites.jl:138-164): structure-of-arrays skeleton of one chain."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class PDMPHistory:
    """Fields and shapes as the reference: ``X``, ``V`` are ``d x n`` (event ``k`` is column ``k``;
    arrays are Fortran-ordered views so each event is contiguous exactly as in Julia), ``t`` etc. have
    length ``n``, ``error_value_ar`` is ``5 x n``, counters are int32 (Composites.jl:145-148)."""
    X: np.ndarray
    V: np.ndarray
    t: np.ndarray
    is_active: np.ndarray
    horizon: np.ndarray
    ar: np.ndarray
    errored_bound: np.ndarray
    error_value_ar: np.ndarray
    rejected: np.ndarray
    hitting_horizon: np.ndarray

    @property
    def x(self):           # Composites.jl:225-234: copies of the columns
        return [self.X[:, k].copy() for k in range(self.X.shape[1])]

    @property
    def v(self):
        return [self.V[:, k].copy() for k in range(self.V.shape[1])]

    def __len__(self):
        return self.t.shape[0]


def history_from_ensemble(out: dict, c: int) -> PDMPHistory:
    """View chain ``c`` of an ensemble result (arrays shaped (C, n_sk, ...)) as a PDMPHistory."""
    X = out["X"][c].T
    n = out["t"].shape[1]
    zeros_i = np.zeros(n, np.int32)
    return PDMPHistory(
        X=X, V=out["V"][c].T, t=out["t"][c],
        is_active=(out["is_active"][c].T.astype(bool) if "is_active" in out else np.ones(X.shape, dtype=bool)),   # Composites.jl:95,256
        horizon=out.get("horizon", np.zeros((1, n)))[c] if "horizon" in out else np.zeros(n),
        ar=out["ar"][c] if "ar" in out else np.zeros(n),
        errored_bound=out["errored_bound"][c] if "errored_bound" in out else zeros_i,
        error_value_ar=out["error_value_ar"][c].T if "error_value_ar" in out else np.zeros((5, n)),
        rejected=out["rejected"][c] if "rejected" in out else zeros_i,
        hitting_horizon=out["hitting_horizon"][c] if "hitting_horizon" in out else zeros_i,
    )
