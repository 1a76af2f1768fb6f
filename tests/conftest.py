import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker).  Test infrastructure only."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    """The product package; compute calls need the HIP library and a GPU.

    The engine fixes a handle's launch policy (kernel form, tuning numbers: the PDMP_* variables) when the handle is created and never
    reads the environment during a launch.  Tests switch forms with ``monkeypatch.setenv`` -- sometimes between two launches of one
    sampler -- so the fixture makes ``handle()`` re-read the policy whenever the PDMP_* part of the environment has changed since that
    sampler last looked (test infrastructure only; a product caller uses ``sampler.reload_policy()``)."""
    import pdmpflux.jl_amd as P
    from pdmpflux.jl_amd.samplers import AbstractPDMP
    if not getattr(AbstractPDMP, "_policy_follows_env", False):
        plain = AbstractPDMP.handle

        def handle(self):
            snap = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("PDMP_")))
            h = plain(self)
            if getattr(self, "_env_snapshot", snap) != snap:
                self.reload_policy()
            self._env_snapshot = snap
            return h
        AbstractPDMP.handle = handle
        AbstractPDMP._policy_follows_env = True
    return P
