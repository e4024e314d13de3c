"""Second, independent FP64 restatement of the residual blocks in torch (float64 + autograd).

Used ONLY by the tests to pin the C++ oracle (and through it the HIP kernels): the functions below are
written from the mathematical definitions in SURVEY.md Appendix A, not from the oracle's code, and their
Jacobians come from torch.autograd instead of hand derivation / Jets:

  * point, IMU, prior factors: the reference hand-codes Jacobians in the right-multiplicative tangent
    convention  q (+) dtheta = q * (1, dtheta/2)  (projection_factor.cpp:234-281 `check`), so here the pose is
    re-parameterised as  pose(delta) = (p + dp, q * deltaQ(dtheta))  and differentiated at delta = 0;
  * line / VP factors: the reference autodiffs w.r.t. the RAW quaternion scalars and keeps columns 0..5
    (Appendix D1), so here autograd differentiates w.r.t. (px,py,pz,qx,qy,qz,qw) and drops the qw column.
"""
import torch

torch.set_default_dtype(torch.float64)


def skew(v):
    z = torch.zeros((), dtype=v.dtype)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def qmul(a, b):  # (x,y,z,w)
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    n2 = (q * q).sum()
    return torch.stack([-q[0], -q[1], -q[2], q[3]]) / n2


def qR(q):
    """I + 2w[v]x + 2[v]x^2 (exact rotation for unit q; the polynomial Eigen evaluates otherwise)."""
    v, w = q[:3], q[3]
    S = skew(v)
    return torch.eye(3) + 2.0 * w * S + 2.0 * S @ S


def deltaQ(th):
    return torch.cat([th / 2.0, torch.ones(1)])


def pose_plus_raw(pose, d):
    """pose (+) d WITHOUT normalisation (derivative at d=0 equals that of the normalised Plus)."""
    return torch.cat([pose[:3] + d[:3], qmul(pose[3:], deltaQ(d[3:]))])


def jac(fn, x):
    return torch.autograd.functional.jacobian(fn, x, create_graph=False, vectorize=False)


# ------------------------------------------------------------------ a5 point reprojection
def point_residual(pose_i, pose_j, ex, lam, pts_i, pts_j, sqrt_info):
    Ri, Rj, Rc = qR(pose_i[3:]), qR(pose_j[3:]), qR(ex[3:])
    pc_i = pts_i / lam
    p_imu_i = Rc @ pc_i + ex[:3]
    pw = Ri @ p_imu_i + pose_i[:3]
    p_imu_j = Rj.T @ (pw - pose_j[:3])
    pc_j = Rc.T @ (p_imu_j - ex[:3])
    return sqrt_info * (pc_j[:2] / pc_j[2] - pts_j[:2])


def point_jacobian(pose_i, pose_j, ex, lam, pts_i, pts_j, sqrt_info):
    """2 x 19 = [d pose_i (6) | d pose_j (6) | d ex (6) | d lambda] in the tangent convention."""
    z6 = torch.zeros(6)
    Ji = jac(lambda d: point_residual(pose_plus_raw(pose_i, d), pose_j, ex, lam, pts_i, pts_j, sqrt_info), z6)
    Jj = jac(lambda d: point_residual(pose_i, pose_plus_raw(pose_j, d), ex, lam, pts_i, pts_j, sqrt_info), z6)
    Je = jac(lambda d: point_residual(pose_i, pose_j, pose_plus_raw(ex, d), lam, pts_i, pts_j, sqrt_info), z6)
    Jl = jac(lambda l: point_residual(pose_i, pose_j, ex, l, pts_i, pts_j, sqrt_info), lam)
    return torch.cat([Ji, Jj, Je, Jl.reshape(2, 1)], dim=1)


# ------------------------------------------------------------------ a7 / a8 line + vanishing point
def Rx(a):
    c, s = torch.cos(a), torch.sin(a); o, z = torch.ones(()), torch.zeros(())
    return torch.stack([torch.stack([o, z, z]), torch.stack([z, c, -s]), torch.stack([z, s, c])])


def Ry(a):
    c, s = torch.cos(a), torch.sin(a); o, z = torch.ones(()), torch.zeros(())
    return torch.stack([torch.stack([c, z, s]), torch.stack([z, o, z]), torch.stack([-s, z, c])])


def Rz(a):
    c, s = torch.cos(a), torch.sin(a); o, z = torch.ones(()), torch.zeros(())
    return torch.stack([torch.stack([c, -s, z]), torch.stack([s, c, z]), torch.stack([z, z, o])])


def line_in_camera(pose, line, ex):
    U = Rx(line[0]) @ Ry(line[1]) @ Rz(line[2])
    n_w = torch.cos(line[3]) * U[:, 0]
    d_w = torch.sin(line[3]) * U[:, 1]
    R_wc = qR(pose[3:]) @ qR(ex[3:])
    t_wc = qR(pose[3:]) @ ex[:3] + pose[:3]
    t_cw = -R_wc.T @ t_wc
    n_c = R_wc.T @ n_w + skew(t_cw) @ (R_wc.T @ d_w)
    d_c = R_wc.T @ d_w
    return n_c, d_c


def line_residual(pose, line, ex, sp, ep, line_factor):
    n_c, _ = line_in_camera(pose, line, ex)
    l = torch.sqrt(n_c[0] ** 2 + n_c[1] ** 2)
    return line_factor * torch.stack([sp @ n_c, ep @ n_c]) / l


def vp_residual(pose, line, ex, vp, vp_factor):
    _, d_c = line_in_camera(pose, line, ex)
    c = torch.abs(d_c @ vp / (torch.linalg.norm(d_c) * torch.linalg.norm(vp)))
    return vp_factor * torch.acos(c).reshape(1)


def raw_jacobian(fn, pose, line):
    """[d/d(raw pose scalars)[:, :6] | d/d line] -- the reference's autodiff + [I6;0] convention."""
    Jp = jac(lambda p: fn(p, line), pose)[:, :6]
    Jl = jac(lambda l: fn(pose, l), line)
    return torch.cat([Jp, Jl], dim=1)


# ------------------------------------------------------------------ a4 IMU
def imu_residual_raw(blk, G, pose_i, sb_i, pose_j, sb_j):
    """15 raw (un-whitened) residuals; blk: dict of torch tensors (sum_dt, delta_p, delta_q, delta_v, lin_ba, lin_bg, jacobian)."""
    dt = blk["sum_dt"]; J = blk["jacobian"]
    Pi, Qi, Pj, Qj = pose_i[:3], pose_i[3:], pose_j[:3], pose_j[3:]
    Vi, Bai, Bgi = sb_i[:3], sb_i[3:6], sb_i[6:9]
    Vj, Baj, Bgj = sb_j[:3], sb_j[3:6], sb_j[6:9]
    dba, dbg = Bai - blk["linearized_ba"], Bgi - blk["linearized_bg"]
    cq = qmul(blk["delta_q"], deltaQ(J[3:6, 12:15] @ dbg))
    cv = blk["delta_v"] + J[6:9, 9:12] @ dba + J[6:9, 12:15] @ dbg
    cp = blk["delta_p"] + J[0:3, 9:12] @ dba + J[0:3, 12:15] @ dbg
    RiT = qR(qinv(Qi))
    rp = RiT @ (0.5 * G * dt * dt + Pj - Pi - Vi * dt) - cp
    rq = 2.0 * qmul(qinv(cq), qmul(qinv(Qi), Qj))[:3]
    rv = RiT @ (G * dt + Vj - Vi) - cv
    return torch.cat([rp, rq, rv, Baj - Bai, Bgj - Bgi])
