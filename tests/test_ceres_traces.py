"""Traces of REAL Ceres (tools/ceres_harness) against the oracle, for every (window, trace) pair committed under tests/golden/ceres/.
None exist yet -- neither Ceres nor Eigen is in the build image -- so today this test is skipped and parity stays UNPINNED by the
reference; the moment a machine with Ceres <= 2.1 produces a pair, the pin is one `git add` away."""
import glob
import os
import struct

import numpy as np
import pytest

from helpers import abi, pose_deltas

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ceres")
PAIRS = [(p, p[:-4] + ".bin") for p in sorted(glob.glob(os.path.join(DIR, "*.trc"))) if os.path.exists(p[:-4] + ".bin")]


def read_trace(path):
    buf = open(path, "rb").read()
    assert buf[:8] == b"UVSTRC01"
    n_it, term, n_pts, n_lines = struct.unpack_from("<4i", buf, 8)
    off, its = 24, []
    for _ in range(n_it):
        d = struct.unpack_from("<6d2i", buf, off); off += 56
        its.append(dict(cost=d[0], cost_change=d[1], radius=d[2], rho=d[3], step_norm=d[4], gmax=d[5], successful=d[6], valid=d[7]))
    tail = np.frombuffer(buf, "<f8", count=176 + n_pts + 4 * n_lines, offset=off)
    return its, term, tail[:77].reshape(11, 7), tail[77:176].reshape(11, 9), tail[176:176 + n_pts], tail[176 + n_pts:].reshape(-1, 4)


@pytest.mark.skipif(not PAIRS, reason="no Ceres traces committed (tools/ceres_harness/README.md): parity unpinned by the reference")
@pytest.mark.parametrize("trace,window", PAIRS)
def test_oracle_follows_ceres(oracle, trace, window):
    its, term, pose, sb, invd, lines = read_trace(trace)
    w = abi.Window.load(window)
    st, rep = oracle.solve(w)
    n = rep.num_iterations
    assert n == len(its) - 1
    # Ceres reports, per iteration k >= 1, the radius AFTER the step and the cost of the current point
    assert [int(i["successful"]) for i in its[1:]] == [1 if a == 1 else 0 for a in rep.accepted[1:n + 1]]
    assert np.allclose([i["cost"] for i in its], np.array(rep.cost[:n + 1]), rtol=1e-6)
    assert np.allclose([i["radius"] for i in its], np.array(rep.radius[:n + 1]), rtol=1e-6)
    dp, dq = pose_deltas(pose, st.pose)
    assert dp < 1e-4 and dq < 1e-4 and np.abs(sb - st.speedbias).max() < 1e-4          # north_star tolerance
    assert np.abs(invd - st.inv_depth).max() < 1e-4 * max(1.0, np.abs(invd).max())


@pytest.mark.gpu
@pytest.mark.skipif(not PAIRS, reason="no Ceres traces committed (tools/ceres_harness/make_traces.sh): parity unpinned by the reference")
@pytest.mark.parametrize("trace,window", PAIRS)
@pytest.mark.parametrize("form", ["persistent", "fused"])
def test_hip_solver_follows_ceres(gpu_api, trace, window, form):
    """The same comparison for the HIP solver itself, both single-window forms: real Ceres' per-iteration (cost, radius, successful) and final
    para_* arrays against the kernel, north_star tolerance on the poses."""
    its, term, pose, sb, invd, lines = read_trace(trace)
    w = abi.Window.load(window)
    s = gpu_api.Solver(max_batch=2)
    if form == "fused":
        s.large_comm_init(None); st, rep, _ = s.large_solve_fused(w)
    else:
        st, rep = s.solve(w)
    s.close()
    n = rep.num_iterations
    assert rep.status == 0 and n == len(its) - 1
    assert [int(i["successful"]) for i in its[1:]] == [1 if a == 1 else 0 for a in rep.accepted[1:n + 1]]
    assert np.allclose([i["cost"] for i in its], np.array(rep.cost[:n + 1]), rtol=1e-6)
    assert np.allclose([i["radius"] for i in its], np.array(rep.radius[:n + 1]), rtol=1e-6)
    dp, dq = pose_deltas(pose, st.pose)
    assert dp < 1e-4 and dq < 1e-4 and np.abs(sb - st.speedbias).max() < 1e-4          # north_star tolerance
    assert np.abs(invd - st.inv_depth).max() < 1e-4 * max(1.0, np.abs(invd).max())
