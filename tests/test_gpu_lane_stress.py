"""GPU: the lane streams of a pipelined batched proof (csrc/batched.hip Pipeline) under stress — 200+ node proofs per configuration with
other legs in between and with the stream -> hardware-queue assignment shifted by streams created before the library's.  Round 2
found one bench.py run in three giving a different one-hot-check proof when the lanes were ordered behind the library stream by events;
round 3 reproduced it (ATLAS_LANE_EVENTS=1, 4 runs of 5), showed that it needs wide lane launches spinning for their challenge (a
one-wavefront gate launch in front of them, channel.hip.h k_ch_gate, removes it: 0 of 8) and keeps both the gate and the host-side
waits.  Also: lanes at T = 2^18 / 2^20, where the waiting grids exceed the resident workgroups (forward progress), must give the
single-stream proof.  The streams created ahead of the library's come from the library's own HIP runtime (a second runtime in the
process, torch's bundled one, cannot open the device once another has).  tools/bisect_lanes.sh runs the same stress under one diagnosis
knob at a time (profiles/r03b_bisect_lanes.txt, r03c_bisect_lanes_nogate.txt)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, env_extra):
    env = dict(os.environ); env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, env=env, capture_output=True, text=True, timeout=900)
    return p.returncode, (p.stdout + p.stderr).strip().splitlines()[-1:]


@pytest.mark.parametrize("env", [{}, {"ATLAS_STRESS_PRE_STREAMS": "1"}, {"ATLAS_STRESS_PRE_STREAMS": "3"}, {"ATLAS_LANE_EVENTS": "1", "ATLAS_STRESS_PRE_STREAMS": "2"}])
def test_lane_streams_deterministic_under_stress(atlas, env):
    rc, tail = _run("stress_lanes.py", ["70"], env)            # 70 repetitions x (ReLU node + Einsum node) x 4 configurations = 560 batched proofs
    assert rc == 0, tail


def test_big_lanes_match_single_stream(atlas):
    rc1, a = _run("stress_big_lanes.py", [], {})
    rc2, b = _run("stress_big_lanes.py", [], {"ATLAS_NO_LANE_STREAMS": "1"})
    assert rc1 == 0 and rc2 == 0, (a, b)
    import re
    sa, sb = re.findall(r"'([0-9a-f]{16})'", a[0]), re.findall(r"'([0-9a-f]{16})'", b[0])
    assert sa and sa == sb, (a, b)
    assert "(1," in a[0] and "(2," not in a[0] and "(3," not in a[0], a
