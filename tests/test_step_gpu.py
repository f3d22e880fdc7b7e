"""`-m gpu`: full NEMARModel.optimize_parameters() on the MI355X kernels vs the CPU oracle and vs the golden
fixtures recorded from the reference — losses, warped images, regulariser, gradients and post-Adam weights.
Tolerances (fp32, different summation order than MKLDNN): losses 2e-4 relative, images 2e-3 abs (values in [-1,1]),
per-tensor gradients 5e-3 relative to the tensor's max, <2% of weights taking a different Adam sign step."""
import pytest

from step_configs import STEP_CONFIGS
import step_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_step_parity(name):
    """Every row of step_parity.run must hold.  One family of rows gets a second, independent attempt: the discriminator
    gradient on identical inputs compares two fp32 implementations across LeakyReLU / InstanceNorm, and roughly one run
    in twenty a pre-activation of these tiny test discriminators (ndf = 8) lies within rounding distance of zero, so the
    two implementations take different slopes there and every gradient below that layer moves by 1e-3..1e-2 of its max
    (measured with tools/flaky_step.py; which run it hits changes because the previous step's weight gradient is summed
    with atomics, i.e. the weights differ in the last bit from run to run).  A systematic error fails both attempts;
    any other row failing fails the test immediately."""
    rows = step_parity.run(name, check=False)
    bad = [r for r in rows if not r[3]]
    if bad and all('grad/D on identical fakes' in r[0] for r in bad):
        rows = step_parity.run(name, check=False)
        bad = [r for r in rows if not r[3]]
    assert not bad, bad[:5]


@pytest.mark.parametrize("name", ["affine128", "unet256"])
def test_batched_passes_match_reference_call_order(name, monkeypatch):
    """T([a ; R(a)]) / D([real ; fake_TR ; fake_RT]) as single batches (NEMAR_BATCHED_PASSES=1) against the reference's
    separate calls (the default): same losses, same gradients up to fp32 summation order."""
    import torch
    import seeded
    cfg = STEP_CONFIGS[name]
    results = []
    for flag in ("1", "0"):
        monkeypatch.setenv("NEMAR_BATCHED_PASSES", flag)
        m = step_parity.build_hip_model(name)
        assert m._batched == (flag == "1")
        a, b = seeded.seeded_images(cfg['batch'], 3, cfg['size'], cfg['size'], cfg['seed'])
        m.set_input({'A': torch.from_numpy(a), 'B': torch.from_numpy(b), 'A_paths': [''], 'B_paths': ['']})
        m.forward()
        m.set_requires_grad([m.netT, m.netR], False)
        m.optimizer_D.zero_grad()
        m.backward_D()
        gd = m.optimizer_D.flat_g.detach().cpu().clone()
        m.set_requires_grad([m.netT, m.netR], True)
        m.set_requires_grad([m.netD, *m.netD_multiresolution], False)
        m.optimizer_R.zero_grad(); m.optimizer_T.zero_grad()
        m.backward_T_and_R()
        results.append(dict(losses=m.get_current_losses(), gd=gd, gt=m.optimizer_T.flat_g.detach().cpu().clone(),
                            gr=m.optimizer_R.flat_g.detach().cpu().clone(), tr=m.fake_TR_B.detach().cpu().clone(),
                            rt=m.fake_RT_B.detach().cpu().clone()))
    x, y = results
    for k in x['losses']:
        assert abs(x['losses'][k] - y['losses'][k]) <= 2e-5 * max(1.0, abs(y['losses'][k])), (k, x['losses'][k], y['losses'][k])
    assert (x['tr'] - y['tr']).abs().max() < 1e-5 and (x['rt'] - y['rt']).abs().max() < 1e-5
    for g in ('gd', 'gt', 'gr'):
        scale = y[g].abs().max().item()
        assert (x[g] - y[g]).abs().max().item() <= 2e-3 * scale, (g, (x[g] - y[g]).abs().max().item(), scale)
