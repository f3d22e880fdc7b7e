"""Step-level parity harness: the MI355X build's NEMARModel vs the CPU oracle (oracle/torch_ref.py) and vs the
golden fixtures recorded from the reference, on identical seeded weights and inputs."""
import os

import numpy as np
import torch

import seeded
from step_configs import STEP_CONFIGS, make_opt
from test_oracle_golden import GOLD, build_ref_model

LR = 2e-4


def load_seeded_into(net, seed, overrides):
    sd = net.state_dict()
    new = seeded.seeded_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, overrides)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})


def build_hip_model(name):
    from nemar_amd.models import create_model
    cfg = STEP_CONFIGS[name]
    opt = make_opt(cfg, gpu_ids=[0])
    m = create_model(opt)
    m.setup(opt)
    load_seeded_into(m.netT, cfg['seed'] + 1, cfg.get('overrides_T'))
    load_seeded_into(m.netR, cfg['seed'] + 2, cfg.get('overrides_R'))
    load_seeded_into(m.netD, cfg['seed'] + 3, cfg.get('overrides_D'))
    for i, d in enumerate(m.netD_multiresolution):
        load_seeded_into(d, cfg['seed'] + 10 + i, cfg.get('overrides_D'))
    return m


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def run(name, report=None, check=True):
    """Returns a list of (what, error, tolerance, ok)."""
    cfg = STEP_CONFIGS[name]
    g = np.load(os.path.join(GOLD, 'step_%s.npz' % name))
    ref = build_ref_model(name)
    hip = build_hip_model(name)
    A, B = seeded.seeded_images(cfg['batch'], 3, cfg['size'], cfg['size'], cfg['seed'])
    tA, tB = torch.from_numpy(A), torch.from_numpy(B)
    rows = []

    def add(what, err, tol):
        rows.append((what, err, tol, bool(err <= tol)))

    for step in range(cfg.get('steps', 1)):
        pre = 's%d/' % step
        ref_losses = ref.optimize_parameters(tA, tB)
        hip.set_input({'A': tA, 'B': tB, 'A_paths': ['a'], 'B_paths': ['b']})
        # capture gradients before Adam consumes them: run the step pieces in the reference's order
        hip.forward()
        hip.set_requires_grad([hip.netT, hip.netR], False)
        hip.optimizer_D.zero_grad()
        hip.backward_D()
        gD = {k: p.grad.detach().cpu().numpy().copy() for k, p in hip.netD.named_parameters()}
        hip.optimizer_D.step()
        hip.set_requires_grad([hip.netT, hip.netR], True)
        hip.set_requires_grad([hip.netD, *hip.netD_multiresolution], False)
        hip.optimizer_R.zero_grad()
        hip.optimizer_T.zero_grad()
        hip.backward_T_and_R()
        gT = {k: p.grad.detach().cpu().numpy().copy() for k, p in hip.netT.named_parameters()}
        gR = {k: p.grad.detach().cpu().numpy().copy() for k, p in hip.netR.named_parameters()}
        hip.optimizer_R.step()
        hip.optimizer_T.step()
        hip.set_requires_grad([hip.netD, *hip.netD_multiresolution], True)
        torch.cuda.synchronize()
        losses = hip.get_current_losses()
        for k in ref_losses:
            tol = 2e-4 * max(1.0, abs(ref_losses[k])) * (step + 1)
            add(pre + 'loss/%s vs oracle' % k, abs(losses[k] - ref_losses[k]), tol)
            add(pre + 'loss/%s vs reference' % k, abs(losses[k] - float(g[pre + 'loss/' + k])), tol)
        add(pre + 'reg vs reference', abs(float(hip.stn_reg_term) - float(g[pre + 'reg'])), 1e-4 * max(1, abs(float(g[pre + 'reg']))))
        for nm in ('fake_B', 'registered_real_A', 'fake_TR_B', 'fake_RT_B'):
            t = getattr(hip, nm).detach().cpu()
            add(pre + 'image/%s vs oracle (max abs)' % nm, float((t - getattr(ref, nm).detach()).abs().max()), 2e-3 * (step + 1))
            add(pre + 'image/%s crop vs reference' % nm, float(np.abs(t[:, :, :16, :16].numpy() - g[pre + 'crop/' + nm]).max()), 2e-3 * (step + 1))
            add(pre + 'image/%s mean vs reference' % nm, abs(t.double().mean().item() - float(g[pre + 'mean/' + nm])), 2e-5 * (step + 1))
        # gradients: relative max-abs error per tensor, for tensors whose oracle gradient is not numerically null
        for nm, mine, theirs in (('T', gT, ref.grads_T), ('R', gR, ref.grads_R), ('D', gD, ref.grads_D)):
            gmax = max(float(v.abs().max()) for v in theirs.values())
            worst, worst_k = 0.0, None
            for k, v in theirs.items():
                vmax = float(v.abs().max())
                if vmax < 1e-5 * gmax:
                    continue                      # e.g. conv biases in front of InstanceNorm: exactly-zero gradient + noise
                e = _rel(mine[k], v.numpy())
                if e > worst:
                    worst, worst_k = e, k
            add(pre + 'grad/%s worst tensor rel err (%s)' % (nm, worst_k), worst, 5e-3 * (step + 1))
        # post-Adam weights: fraction of elements whose update differs by more than half a step
        for nm, net, theirs in (('T', hip.netT, ref.T), ('R', hip.netR, ref.R), ('D', hip.netD, ref.D)):
            bad = tot = 0
            for k, p in net.named_parameters():
                if not k.endswith('weight'):
                    continue
                d = (p.detach().cpu() - theirs[k].detach()).abs()
                bad += int((d > 0.5 * LR * (step + 1)).sum())
                tot += d.numel()
            add(pre + 'adam/%s fraction of weights off by > lr/2' % nm, bad / max(tot, 1), 0.02 * (step + 1))
    if report:
        with open(report, 'a') as f:
            f.write('== %s\n' % name)
            for r in rows:
                f.write('%-70s err=%.3e tol=%.1e %s\n' % (r[0], r[1], r[2], 'ok' if r[3] else 'FAIL'))
    if check:
        bad = [r for r in rows if not r[3]]
        assert not bad, bad[:5]
    return rows


if __name__ == '__main__':
    import sys
    out = sys.argv[1] if len(sys.argv) > 1 else '/dev/stdout'
    for name in STEP_CONFIGS:
        try:
            run(name, report=out, check=False)
        except Exception as e:  # noqa
            import traceback
            with open(out, 'a') as f:
                f.write('== %s EXCEPTION\n%s\n' % (name, traceback.format_exc()))
