"""Per-tensor gradient comparison of each network (HIP build vs CPU oracle) on a simple loss."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import seeded
from step_configs import STEP_CONFIGS, make_opt
from oracle import torch_ref as R
from nemar_amd.models import networks, stn
from nemar_amd import ops

def seeded_load(net, seed, ov=None):
    sd = net.state_dict()
    new = seeded.seeded_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, ov)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
    return {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in new.items()}

def report(name, net, P, extra=None):
    gmax = max(float(p.grad.abs().max()) for p in P.values() if p.grad is not None)
    for k, p in net.named_parameters():
        g = p.grad.detach().cpu()
        r = P[k].grad
        if r is None: continue
        e = float((g - r).abs().max()); m = float(r.abs().max())
        print('%-3s %-60s max|ref|=%.3e  err=%.3e  rel=%.3e %s' % (name, k, m, e, e / (m + 1e-30), '' if m > 1e-5 * gmax else '(null grad)'))

dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = STEP_CONFIGS['affine128']; opt = make_opt(cfg, [0])
N, S = 2, 128
A, B = seeded.seeded_images(N, 3, S, S, 5)
tA, tB = torch.from_numpy(A), torch.from_numpy(B)
# ---- D
netD = networks.define_D(6, 8, 'basic', 3, 'instance', 'normal', 0.02, [0])
PD = seeded_load(netD, 3)
opt_D = ops.FlatAdam(netD.parameters())
out = netD(tA.to(dev), tB.to(dev))
w = torch.from_numpy(seeded.uniform(tuple(out.shape), 77)).to(dev)
ref = R.nlayer_discriminator(PD, torch.cat([tA, tB], 1))
print('D fwd err', float((out.detach().cpu() - ref.detach()).abs().max()))
torch.autograd.backward([out], [w]); (ref * w.cpu()).sum().backward()
report('D', netD, PD)
# ---- T
netT = networks.define_G(3, 3, 8, 'resnet_3blocks', 'instance', False, 'normal', 0.02, [0])
PT = seeded_load(netT, 1)
opt_T = ops.FlatAdam(netT.parameters())
x = tA.to(dev).requires_grad_(True); xr = tA.clone().requires_grad_(True)
out = netT(x); ref = R.resnet_generator(PT, xr, 3)
w = torch.from_numpy(seeded.uniform(tuple(out.shape), 78)).to(dev)
print('T fwd err', float((out.detach().cpu() - ref.detach()).abs().max()))
torch.autograd.backward([out], [w]); (ref * w.cpu()).sum().backward()
report('T', netT, PT)
print('T dx rel', float((x.grad.cpu() - xr.grad).abs().max() / xr.grad.abs().max()))

# ---- D step structure: three passes, BCE, multi-root backward
print('--- D step structure')
netD2 = networks.define_D(6, 8, 'basic', 3, 'instance', 'normal', 0.02, [0])
PD2 = seeded_load(netD2, 3)
optD2 = ops.FlatAdam(netD2.parameters())
f1, f2 = seeded.seeded_images(N, 3, S, S, 9)
tf1, tf2 = torch.from_numpy(f1), torch.from_numpy(f2)
crit = networks.GANLoss('vanilla')
one = torch.ones((), device=dev)
dA, dB, d1, d2 = tA.to(dev), tB.to(dev), tf1.to(dev), tf2.to(dev)
for npass in (1, 2, 3):
    optD2.zero_grad()
    for p in PD2.values(): p.grad = None
    imgs = [(dB, tB, True), (d1, tf1, False), (d2, tf2, False)][:npass]
    terms = [crit(netD2(dA, di.detach()), real, 0.5) for di, _, real in imgs]
    torch.autograd.backward(terms, [one] * len(terms))
    loss = sum(0.5 * R.gan_loss(R.nlayer_discriminator(PD2, torch.cat([tA, ti], 1)), real) for _, ti, real in imgs)
    loss.backward()
    print('npass', npass, 'loss err', abs(float(sum(float(t) for t in terms)) - float(loss)))
    report('D%d' % npass, netD2, PD2)
