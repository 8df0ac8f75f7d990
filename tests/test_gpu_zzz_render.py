"""The n=2 "render" generator (the default materialised generator since round 3) and the summing whole-line writer
(THETA_N2_ENUM_RENDER=0) against the lane-stream generator (THETA_N2_ENUM_LEGACY=1) on the device."""
import pytest

pytestmark = pytest.mark.gpu

_RENDER_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import theta_amd
m, k = int(sys.argv[2]), int(sys.argv[3])
ctx = theta_amd.default_context()
p = theta_amd.Problem(ctx, 2, m, 2, [1] * m, [1] * m, [0] * m, [k] * m)
cnt = int(min(p.count, 3_000_000))
for b, c in ((0, cnt), (p.count // 3, min(cnt, 1_000_001, p.count - p.count // 3)), (max(0, p.count - 777_777), min(p.count, 777_777))):
    if c < 1:
        continue
    os.environ["THETA_N2_ENUM_LEGACY"] = "1"
    old = p.enumerate(b, c)
    del os.environ["THETA_N2_ENUM_LEGACY"]
    new = p.enumerate(b, c)                                  # the default: the render kernel
    os.environ["THETA_N2_ENUM_RENDER"] = "0"
    lines = p.enumerate(b, c)                                # the summing whole-line writer
    del os.environ["THETA_N2_ENUM_RENDER"]
    if not np.array_equal(lines, old):
        print("MISMATCH (summing writer) m=%d k=%d range (%d, %d)" % (m, k, b, c))
        sys.exit(4)
    if not np.array_equal(new, old):
        bad = int(np.nonzero((new != old).any(axis=1))[0][0])
        print("MISMATCH m=%d k=%d range (%d, %d): first differing record %d" % (m, k, b, c, bad))
        sys.exit(3)
print("equal")
"""


@pytest.mark.parametrize("m,k", [(50, 6), (100, 5), (25, 5), (7, 3), (64, 9), (130, 2)])
def test_n2_render_generator_equals_the_lane_stream_generator(m, k):
    """n2_enumerate_render_kernel (records by scatter + prefix sum, verified lane by lane on the CPU in
    tests/test_n2_render_cpu.py) against the one-stream-per-lane kernel, whole ranges and ragged sub-ranges.  (Round 2's
    version of this test asked for ranges beyond the end of small spaces -- "rank range out of bounds" on m = 25, 7, 130 -- and
    read as a kernel failure; the kernel never differed.)  In a child process, so that the environment switches stay local."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "-c", _RENDER_CHILD, ROOT, str(m), str(k)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "equal" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
