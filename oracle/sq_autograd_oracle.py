"""TEST INFRASTRUCTURE -- never imported by the product (odam_amd/); only tests/ and bench.py's cpu_baseline leg use it.

The reference's super-quadric fit AS THE REFERENCE RUNS IT: torch CPU tensors, reverse-mode autograd, torch.optim.Adam,
200 steps (likojack/ODAM src/super_quadric/sq_libs.py:432-475, the `run` loop), restated so that it can be TIMED on a box where
/root/reference does not exist.  oracle/sq_oracle.c is the other restatement: closed-form gradient, plain C, an order of magnitude
faster than this loop -- which is why a CPU baseline composed from it flatters the reference.

What follows what:
    squashing                       sq_libs.py:26-27
    surface points                  learnable_primitives/sampling.py:558-622 (sample_points_on_surface; the normals it also returns
                                    are computed and dropped there and feed nothing, they are left out)
    the sampler behind it           fast_sampler/sampling.cpp:59-215 -> oracle/sq_oracle.c::oracle_sq_sample (pinned bit for bit against
                                    the compiled reference source in oracle/_ref)
    rotz, compute_ellipsoid_points  sq_libs.py:545-595
    constraint_2d                   sq_libs.py:395-430
    Adam groups, learning rates     sq_libs.py:373-388 (translate | angle | scales at 0.01, shapes at 0.1)
    scale prior, x20                sq_libs.py:461-463
    set_detect_anomaly(True)        sq_libs.py:456 -- on every step in the reference; `anomaly` switches it (both are timed)

Pinned by tests/test_sq_oracle.py::test_autograd_restatement_is_the_reference_fit against tests/golden/sq_survey.npz (final
parameters of the reference's own run of the same problems).
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _sampler():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(os.path.join(_HERE, "libodam_oracle.so"))
    return _LIB


def _sample(scales_sq, eps):
    """etas, omegas [1000] for one primitive (EqualDistanceSamplerSQ(1000).sample_on_batch, sampling.py:455-470)"""
    fp = ctypes.POINTER(ctypes.c_float)
    a = np.ascontiguousarray(scales_sq, np.float32)
    e = np.ascontiguousarray(eps, np.float32)
    et, om = np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    _sampler().oracle_sq_sample(a.ctypes.data_as(fp), e.ctypes.data_as(fp), et.ctypes.data_as(fp), om.ctypes.data_as(fp))
    return et, om


def _fexp(x, p):
    return torch.sign(x) * (torch.abs(x) ** p)


def _points(translate, angle, scales, shapes):
    """compute_ellipsoid_points(use_numpy=False), sq_libs.py:574-595"""
    cosz, sinz = torch.cos(angle), torch.sin(angle)
    zeros = angle.detach() * 0
    ones = zeros.detach() + 1
    R = torch.stack([cosz, -sinz, zeros, sinz, cosz, zeros, zeros, zeros, ones], dim=0).reshape(3, 3)
    a = (scales ** 2).unsqueeze(0).unsqueeze(0)
    e = (torch.sigmoid(shapes) * (1.6 - 0.2) + 0.2).unsqueeze(0).unsqueeze(0)
    etas, omegas = _sample(a.detach().numpy().reshape(3), e.detach().numpy().reshape(2))
    etas[etas == 0] += 1e-6
    omegas[omegas == 0] += 1e-6
    etas = a.new_tensor(etas.reshape(1, 1, -1))
    omegas = a.new_tensor(omegas.reshape(1, 1, -1))
    a1, a2, a3 = (a[:, :, i].unsqueeze(-1) for i in range(3))
    e1, e2 = (e[:, :, i].unsqueeze(-1) for i in range(2))
    x = a1 * _fexp(torch.cos(etas), e1) * _fexp(torch.cos(omegas), e2)
    y = a2 * _fexp(torch.cos(etas), e1) * _fexp(torch.sin(omegas), e2)
    z = a3 * _fexp(torch.sin(etas), e1)
    x = ((x > 0).float() * 2 - 1) * torch.max(torch.abs(x), x.new_tensor(1e-6))
    y = ((y > 0).float() * 2 - 1) * torch.max(torch.abs(y), x.new_tensor(1e-6))
    z = ((z > 0).float() * 2 - 1) * torch.max(torch.abs(z), x.new_tensor(1e-6))
    pts = torch.stack([x, y, z], -1)[0, 0]
    pts = pts @ R.T
    pts += translate.unsqueeze(0)
    return pts


def _constraint_2d(pts_w, Ms, tgt, mask):
    """sq_libs.py:395-430; tgt[:, d] = -gt_line[name][-1] is what the L1 compares with (the target array of odam_amd.sq.lines_to_targets
    already carries that sign), mask[:, d] = 1 where the detection has that edge"""
    n_frames = Ms.shape[0]
    pts_w = pts_w.repeat(n_frames, 1, 1)
    # geometry_utils.py:25-39 (get_homogeneous of a 3-D tensor): the FIRST copy of the repeated points gets the ones column and goes
    # on as [1, N, 4]; the product with the F matrices broadcasts it -- so autograd sums the F views' contributions inside the matmul's
    # backward, and the other F - 1 copies of the repeat receive no gradient.  Restated as such: the summation order is part of the bits.
    first = pts_w[0]
    hom = torch.cat([first, torch.ones_like(first[:, 2:])], dim=1)[None, :, :]
    pixels = hom @ Ms.permute(0, 2, 1)
    valid = pixels[:, :, 2] > 0.5
    pixels = pixels[:, :, :2] / (torch.abs(pixels[:, :, 2:]) + 1e-6)
    big = torch.ones_like(pixels[:, :, 0]) * 1000000
    pred = [torch.min(torch.where(valid, pixels[:, :, 0], big), dim=1).values,
            torch.max(torch.where(valid, pixels[:, :, 0], -big), dim=1).values,
            torch.min(torch.where(valid, pixels[:, :, 1], big), dim=1).values,
            torch.max(torch.where(valid, pixels[:, :, 1], -big), dim=1).values]
    loss = 0
    for d in range(4):
        l = torch.nn.functional.l1_loss(pred[d], tgt[:, d], reduction="none")
        l = torch.where(torch.isnan(l), torch.zeros_like(l), l)
        l = l * mask[:, d]
        loss = loss + torch.mean(l)
    return loss


def fit(p0, P, tgt, mask, icov=None, n_iters=200, optimise_shapes=True, anomaly=True):
    """One object's fit, the reference's way.  p0 [9] = translate(3) | angle | scales(3) | shapes(2) (odam_amd.sq.init_params layout),
    P [F, 12] projection rows, tgt / mask [F, 4] (x_min, x_max, y_min, y_max), icov [3, 3] inverse scale covariance of the object's class
    or None (prior off).  Returns the nine parameters after n_iters Adam steps, float32."""
    p0 = np.asarray(p0, np.float32)
    translate = torch.tensor(p0[0:3], dtype=torch.float32, requires_grad=True)
    angle = torch.tensor(p0[3], dtype=torch.float32, requires_grad=True)
    scales = torch.tensor(p0[4:7], dtype=torch.float32, requires_grad=True)
    shapes = torch.tensor(p0[7:9], dtype=torch.float32, requires_grad=True)
    groups = [{"params": [translate, angle, scales]}]
    if optimise_shapes:
        groups.append({"params": [shapes], "lr": 0.1})
    opt = torch.optim.Adam(groups, lr=0.01)
    Ms = torch.tensor(np.asarray(P, np.float32).reshape(-1, 3, 4))
    tgt_t = torch.tensor(np.asarray(tgt, np.float32))
    mask_t = torch.tensor(np.asarray(mask, np.float32))
    icov_t = None if icov is None else torch.tensor(np.asarray(icov, np.float32).reshape(3, 3))
    scales_init = scales.detach().clone()
    was = torch.is_anomaly_enabled()
    try:
        for _ in range(n_iters):
            torch.autograd.set_detect_anomaly(bool(anomaly))
            opt.zero_grad()
            pts_w = _points(translate, angle, scales, shapes)
            loss = _constraint_2d(pts_w, Ms, tgt_t, mask_t)
            if icov_t is not None:
                d = (scales_init - scales)[None, :]
                loss = loss + (d @ icov_t @ d.T)[0, 0] * 20
            loss.backward()
            opt.step()
    finally:
        torch.autograd.set_detect_anomaly(was)
    return np.concatenate([translate.detach().numpy(), angle.detach().numpy().reshape(1), scales.detach().numpy(),
                           shapes.detach().numpy()]).astype(np.float32)
