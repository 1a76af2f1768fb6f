"""The checker's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer ("sanitizers on the CPU build"; GPU
sanitizers are not available on the pool).  `make -C oracle check-sanitize` builds oracle/sanitize_driver.c with
-fsanitize=address,undefined -fno-sanitize-recover=all and runs 153 cases: 131 sampler / target / order / slope-mode cases plus the callers of the
path (the three sample_from_skeleton methods with ragged histories, the time-horizon variant incl. a capacity that is too small);
any finding aborts the driver.  CPU only, a few seconds."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="no C compiler")
def test_checker_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "check-sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "0 failed" in r.stdout
