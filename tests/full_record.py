"""What a full-width step is reduced to (scalars + small crops), shared by the fixture generator — which runs the
REFERENCE's NEMARModel (tests/golden/make_golden.py, build container only) — and by the `-m gpu` test, which runs the
MI355X build's NEMARModel through the very same function (tests/test_step_full_gpu.py)."""
import numpy as np
import torch

import seeded


def proj(t, seed, stream):
    """<t, r> / sqrt(numel) with r the seeded U[-1,1) vector (seed, stream): one scalar that moves with every element."""
    r = torch.from_numpy(seeded.uniform((t.numel(),), seed, stream)).double()
    return float((t.detach().double().cpu().reshape(-1) * r).sum().item() / np.sqrt(t.numel()))


def full_step_record(m, A, B, seed):
    """One optimize_parameters() of `m` (anything with NEMARModel's attributes) on the numpy batch (A, B)."""
    out = {}
    p0 = next(m.netT.parameters())
    a, b = torch.from_numpy(A).to(p0.device, p0.dtype), torch.from_numpy(B).to(p0.device, p0.dtype)
    if hasattr(m.netR, 'offset_map'):            # (the affine STN has no dense field: its sampling grid is recorded instead)
        with torch.no_grad():
            off = m.netR.offset_map(a, b)        # the deformation field, before the update
        out['offsets/mean'], out['offsets/absmean'] = off.double().mean().item(), off.double().abs().mean().item()
        out['offsets/proj'] = proj(off, seed, 900)
        out['offsets/cropc'] = _cropc(off)
    else:
        with torch.no_grad():
            grid = m.netR.get_grid(a, b).permute(0, 3, 1, 2)
        out['offsets/mean'], out['offsets/absmean'] = grid.double().mean().item(), grid.double().abs().mean().item()
        out['offsets/proj'] = proj(grid, seed, 900)
        out['offsets/cropc'] = _cropc(grid)
    m.set_input({'A': a, 'B': b, 'A_paths': ['a'], 'B_paths': ['b']})
    m.optimize_parameters()
    for k, v in m.get_current_losses().items():
        out['loss/' + k] = float(v)
    out['reg'] = float(m.stn_reg_term)
    for i, nm in enumerate(('fake_B', 'registered_real_A', 'fake_TR_B', 'fake_RT_B')):
        t = getattr(m, nm).detach()
        out['mean/' + nm], out['absmean/' + nm] = t.double().mean().item(), t.double().abs().mean().item()
        out['proj/' + nm] = proj(t, seed, 901 + i)
        out['crop0/' + nm] = t[:, :, :8, :8].double().cpu().numpy().copy()
        out['cropc/' + nm] = _cropc(t)
    nets = [('T', m.netT), ('R', m.netR), ('D', m.netD)] + [('Dmr%d' % i, d) for i, d in enumerate(m.netD_multiresolution)]
    for nm, net in nets:
        for j, (k, p) in enumerate(net.named_parameters()):
            if p.grad is not None:
                out['gradnorm/%s/%s' % (nm, k)] = p.grad.double().norm().item()
                out['gradproj/%s/%s' % (nm, k)] = proj(p.grad, seed, 1000 + j)
                out['gradmax/%s/%s' % (nm, k)] = p.grad.double().abs().max().item()
            out['psum/%s/%s' % (nm, k)] = p.detach().double().sum().item()
            out['pabs/%s/%s' % (nm, k)] = p.detach().double().abs().sum().item()
    return out


def _cropc(t):
    h, w = t.shape[2] // 2, t.shape[3] // 2
    return t[:, :, h - 4:h + 4, w - 4:w + 4].double().cpu().numpy().copy()


def registration_record(netR, l1, A, B, seed, lambda_recon=100.0, lambda_smooth=10.0):
    """The registration SUB-MODEL of a step — deformation field, warp of A, smoothness term, and the gradients of
    lambda_recon * L1(warp(A), B) + lambda_smooth * reg w.r.t. every parameter of the registration net — reduced to scalars and
    small crops.  Small enough to run the REFERENCE's UnetSTN in fp64 at 1024x1024 (the full step there needs > 60 GB), so BASELINE
    config 5's stress path (large-field grid_sample + smoothness + the deep registration net) has an fp64 truth of its own.
    `l1(a, b, weight)` -> weight * mean|a - b| (torch's L1Loss for the reference, ops.l1_loss for the build)."""
    out = {}
    p0 = next(netR.parameters())
    a, b = torch.from_numpy(A).to(p0.device, p0.dtype), torch.from_numpy(B).to(p0.device, p0.dtype)
    for p in netR.parameters():
        p.grad = None if getattr(p, '_flat_grad', None) is None else p.grad
    warped, reg = netR(a, b, apply_on=[a])
    w = warped[0]
    rec = l1(w, b, lambda_recon)
    out['loss/recon'], out['reg'] = float(rec), float(reg)
    torch.autograd.backward([rec, reg], [torch.ones_like(rec), torch.full_like(reg, lambda_smooth)])
    with torch.no_grad():
        off = netR.offset_map(a, b)
    out['offsets/mean'], out['offsets/absmean'] = off.double().mean().item(), off.double().abs().mean().item()
    out['offsets/proj'] = proj(off, seed, 900)
    out['offsets/cropc'] = _cropc(off)
    t = w.detach()
    out['mean/warped'], out['absmean/warped'] = t.double().mean().item(), t.double().abs().mean().item()
    out['proj/warped'] = proj(t, seed, 901)
    out['crop0/warped'] = t[:, :, :8, :8].double().cpu().numpy().copy()
    out['cropc/warped'] = _cropc(t)
    for j, (k, p) in enumerate(netR.named_parameters()):
        if p.grad is not None:
            out['gradnorm/R/' + k] = p.grad.double().norm().item()
            out['gradproj/R/' + k] = proj(p.grad, seed, 1000 + j)
            out['gradmax/R/' + k] = p.grad.double().abs().max().item()
    return out


def traj_record(m, A, B, seed, steps):
    """`steps` CONSECUTIVE optimize_parameters() of `m` on the same numpy batch (free running: every step starts from the weights and Adam
    moments the previous one left).  Per step: the 8 losses, the regularisation term, statistics + a seeded projection of the
    deformation field the step's forward pass used and of the translated image; after the last step the per-network parameter sums."""
    out = {}
    p0 = next(m.netT.parameters())
    a, b = torch.from_numpy(A).to(p0.device, p0.dtype), torch.from_numpy(B).to(p0.device, p0.dtype)
    for s in range(steps):
        pre = 's%02d/' % s
        with torch.no_grad():
            off = m.netR.offset_map(a, b)
        out[pre + 'offsets/mean'], out[pre + 'offsets/absmean'] = off.double().mean().item(), off.double().abs().mean().item()
        out[pre + 'offsets/proj'] = proj(off, seed, 900)
        m.set_input({'A': a, 'B': b, 'A_paths': ['a'], 'B_paths': ['b']})
        m.optimize_parameters()
        for k, v in m.get_current_losses().items():
            out[pre + 'loss/' + k] = float(v)
        out[pre + 'reg'] = float(m.stn_reg_term)
        t = m.fake_B.detach()
        out[pre + 'absmean/fake_B'] = t.double().abs().mean().item()
        out[pre + 'proj/fake_B'] = proj(t, seed, 901)
    for nm, net in (('T', m.netT), ('R', m.netR), ('D', m.netD)):
        out['final/psum/' + nm] = sum(p.detach().double().sum().item() for p in net.parameters())
        out['final/pabs/' + nm] = sum(p.detach().double().abs().sum().item() for p in net.parameters())
    return out
