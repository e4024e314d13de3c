"""Independent numpy restatement of the Levenberg-Marquardt controller the solver must reproduce (SURVEY.md Appendix B =
Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy with the options of estimator.cpp:982-994).

TEST INFRASTRUCTURE ONLY.  Written from Appendix B, not from oracle/uvs_oracle.cpp: it shares NO solver code with the C++ oracle or
with the HIP kernel -- dense normal equations over the whole parameter vector (no Schur complement), numpy `solve`, Ceres'
Jacobi-SCALED coordinates.  The residual blocks and their Jacobians come from a per-block evaluation callback (`evaluate(window)
-> abi.Eval`); the tests pass the oracle's factor evaluation, which is pinned element-wise against torch autograd in
tests/test_oracle_factors.py.  What this file pins is therefore the CONTROLLER: Jacobi scaling computed once, the clamped Marquardt
diagonal reused after a rejected step, the exact damped solve, model_cost_change, the order tolerance tests -> accept / reject,
the radius rules, the iteration count, and both `function_tol_keeps_candidate` variants.
"""
import copy

import numpy as np

from helpers import abi, dense_normal_equations, param_layout

NF = abi.NUM_FRAMES


def quat_mul(a, b):      # (x, y, z, w)
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def pose_plus(pose, d):
    """PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-18): p + dp, normalize(q * (dtheta / 2, 1))."""
    q = quat_mul(pose[3:], np.array([0.5 * d[3], 0.5 * d[4], 0.5 * d[5], 1.0]))
    return np.concatenate([pose[:3] + d[:3], q / np.linalg.norm(q)])


def plus(w, delta):
    """x (+) delta over [frames: 6 pose tangent + 9 speed/bias | inverse depths | line parameters]; returns a new window."""
    F, Np, Nl, P = param_layout(w)
    c = copy.copy(w)
    c.pose = w.pose.copy(); c.speedbias = w.speedbias.copy()
    for f in range(NF):
        c.pose[f] = pose_plus(w.pose[f], delta[15 * f:15 * f + 6])
        c.speedbias[f] = w.speedbias[f] + delta[15 * f + 6:15 * f + 15]
    c.inv_depth = w.inv_depth + delta[F:F + Np]
    c.line_orth = w.line_orth + delta[F + Np:].reshape(Nl, 4)
    return c


def ambient(w):
    return np.concatenate([w.pose.ravel(), w.speedbias.ravel(), w.inv_depth, w.line_orth.ravel()])


def projected_gradient_max_norm(w, g):
    """|| x - Plus(x, -g) ||_inf (Appendix B.2): manifold blocks through Plus, Euclidean blocks are just |g|."""
    return np.abs(ambient(plus(w, -g)) - ambient(w)).max()


class Trace:
    def __init__(self):
        self.accepted, self.radius, self.cost, self.candidate_cost, self.model_cost_change = [1], [], [], [0.0], [0.0]
        self.termination = abi.TERM_NAMES.index("NO_CONVERGENCE"); self.num_iterations = 0; self.final = None; self.final_cost = None


def solve(w, evaluate, options=None):
    """Returns (final window, Trace).  `options` = abi.Options (defaults: Ceres' defaults + NUM_ITERATIONS = 10)."""
    o = options if options is not None else abi.default_options()
    tr = Trace()
    x = w
    ev = evaluate(x)
    H, g = dense_normal_equations(x, ev)
    cost = ev.cost
    # B.2: Jacobi scaling, once, from the first Jacobian: s_k = 1 / (1 + sqrt(sum_rows J_k^2)) = 1 / (1 + sqrt(H_kk))
    s = 1.0 / (1.0 + np.sqrt(np.diag(H))) if o.jacobi_scaling else np.ones(len(g))
    radius, decrease_factor = o.initial_trust_region_radius, 2.0
    tr.radius.append(radius); tr.cost.append(cost)
    x_norm = np.linalg.norm(ambient(x))
    invalid = 0
    diag = None
    fresh = True              # a new linearization point => recompute the Marquardt diagonal
    it = 0
    while True:
        if it >= o.max_num_iterations: tr.termination = 0; break
        if projected_gradient_max_norm(x, g) <= o.gradient_tolerance: tr.termination = 1; break
        if radius <= o.min_trust_region_radius: tr.termination = 4; break
        it += 1
        Hs, gs = H * np.outer(s, s), g * s                      # (J s)^T (J s), (J s)^T r
        if fresh:
            diag = np.clip(np.diag(Hs), o.min_lm_diagonal, o.max_lm_diagonal)
            fresh = False
        D2 = diag / radius                                       # D^T D, B.3
        y = np.linalg.solve(Hs + np.diag(D2), -gs)
        model_cost_change = -(y @ gs) - 0.5 * (y @ Hs @ y)       # -(J y) . (r + J y / 2)
        tr.model_cost_change.append(model_cost_change)
        if not (np.all(np.isfinite(y)) and model_cost_change > 0.0):     # invalid step
            invalid += 1
            radius /= decrease_factor; decrease_factor *= 2.0
            tr.accepted.append(-1); tr.radius.append(radius); tr.cost.append(cost); tr.candidate_cost.append(cost)
            if invalid >= o.max_consecutive_invalid_steps: tr.termination = 5; break
            continue
        invalid = 0
        delta = s * y
        xc = plus(x, delta)
        evc = evaluate(xc)
        cand = evc.cost if np.isfinite(evc.cost) else np.finfo(float).max
        tr.candidate_cost.append(cand)
        step_norm = np.linalg.norm(ambient(xc) - ambient(x))
        rho = (cost - cand) / model_cost_change
        successful = rho > o.min_relative_decrease
        stop = None
        if step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance): stop = 2          # B.4: tolerance tests BEFORE accept / reject
        elif abs(cost - cand) <= o.function_tolerance * cost: stop = 3
        if stop is not None and not (o.function_tol_keeps_candidate and successful):
            tr.accepted.append(0); tr.radius.append(radius); tr.cost.append(cost)
            tr.termination = stop; break
        if successful:                                            # B.5
            x, cost, ev = xc, cand, evc
            H, g = dense_normal_equations(x, ev)
            x_norm = np.linalg.norm(ambient(x))
            radius = min(o.max_trust_region_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease_factor = 2.0
            fresh = True
            tr.accepted.append(1)
        else:
            radius /= decrease_factor; decrease_factor *= 2.0
            tr.accepted.append(0)
        tr.radius.append(radius); tr.cost.append(cost)
        if stop is not None: tr.termination = stop; break
    tr.num_iterations = it; tr.final = x; tr.final_cost = cost
    return x, tr
