"""init -> work -> shutdown -> init -> work in one process: the arenas the MSM / HyperKZG code keeps between calls are
released by atlas_shutdown and rebuilt afterwards.  Child process: the session fixture must not see a shutdown."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import jolt_atlas_amd as A
from oracle import orc
res = []
for it in range(2):
    A.init(0)
    tau = orc.random_fr(1, 7)[0]
    srs = A.SRS.generate(tau, 1 << 16)
    p = A.MultilinearPolynomial.from_fr(orc.random_fr(1 << 16, 8))
    pt = [3 * i + 1 for i in range(16)]
    com, w, v = A.HyperKZG.open(srs, p, pt, A.Blake2bTranscript(b"life"))
    res.append((np.asarray(v).tobytes(), np.asarray(w).tobytes()))
    p.free(); srs.free()
    assert A.lib.atlas_shutdown() == 0
assert res[0] == res[1]
print("LIFECYCLE_OK")
'''


def test_shutdown_and_reinit():
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LIFECYCLE_OK" in r.stdout, (r.stdout[-300:], r.stderr[-600:])
