"""How sensitive are the LM traces of the STRESS windows (tests/gpu_soak_rejections.py: heavily perturbed, 30 iterations) to round-off alone?
CPU only, no HIP code involved: the same oracle sources built twice -- `-O2 -ffp-contract=off` (the checker of the test suite) and
`-O3 -march=native` (fused multiply-adds, other summation orders where the compiler vectorises) -- solve the same windows, and the script reports the
windows whose traces differ, the drift of the accepted cost up to the first differing decision, and the end states.  If two builds of ONE
sequential program part ways on the same windows and in the same manner as the HIP solver does from the oracle, the divergence is a property of
those windows (exponential amplification of round-off by the LM iteration far from the optimum), not of either solver.
python tests/cpu_soak_oracle_variants.py [n windows]      (windows without a prior only: their generation needs no GPU)"""
import sys, os, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import abi, synth, pose_deltas, first_divergence
from oracle_binding import Oracle, ROOT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = tempfile.mkdtemp()
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "OUT=" + out])
a = Oracle(); b = Oracle(os.path.join(out, "liboracle_native.so"))
opts = abi.default_options(); opts.max_num_iterations = 30
rng = np.random.default_rng(77)
ndiff = 0; rows = []
for i in range(N):
    # the same random stream as gpu_soak_rejections.py (every window draws, only the prior-free ones are solved here)
    w = synth.make_window(9000 + i).copy()
    amp = float(rng.choice([0.05, 0.2, 0.5]))
    w.pose[2:, :3] += amp * rng.standard_normal((9, 3)); w.inv_depth *= np.exp(amp * rng.standard_normal(len(w.inv_depth)))
    w.line_orth += 0.3 * amp * rng.standard_normal(w.line_orth.shape)
    if i % 2: continue
    sa, ra = a.solve(w, opts); sb, rb = b.solve(w, opts)
    d = first_divergence(rb, ra, opts)
    if d is None: continue
    ndiff += 1
    dp, dq = pose_deltas(sb.pose, sa.pose)
    drift = ["%.0e%s" % (abs(rb.cost[q] - ra.cost[q]) / abs(ra.cost[q]), "" if ra.accepted[q] == 1 else "r") for q in range(d["k"] + 1)]
    print("window %2d amp %.2f: first differing decision at k = %d (%s: %s | %s); end states dp %.2e m, dq %.2e rad, final cost %.9g | %.9g" % (
        i, amp, d["k"], d["kind"], d.get("rho_gpu", d["gpu"]), d.get("rho_oracle", d["oracle"]), dp, dq, rb.final_cost, ra.final_cost))
    print("    drift of the accepted cost: " + " ".join(drift))
print("%d prior-free stress windows, two CPU builds of the oracle: LM traces differ in %d" % ((N + 1) // 2, ndiff))
