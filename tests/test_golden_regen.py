"""The pin is reproducible: `python -m oracle.gen_golden` on a clean output directory regenerates every
fixture of tests/golden/ bit for bit from the imported reference (build container only -- the reference
tree does not exist on the GPU box, where this test is skipped)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree (build container only)")
def test_fixtures_regenerate_bit_for_bit(tmp_path):
    env = dict(os.environ, ESR_GOLDEN_OUT=str(tmp_path), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "oracle.gen_golden"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    fresh = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert fresh == committed, (set(fresh) ^ set(committed))
    for f in committed:
        with np.load(os.path.join(GOLDEN, f), allow_pickle=False) as a, np.load(tmp_path / f, allow_pickle=False) as b:
            assert sorted(a.files) == sorted(b.files), f
            for k in a.files:
                x, y = a[k], b[k]
                assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), (f, k)
