"""One ResnetBlock of the wide route as the step runs it (ops._ResBlock: 256 channels, 64 x 64, batch 16, dropout on), forward + backward,
repeated — for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_run.sh ... "pmc:tools/pmc_block.py") and --kernel-trace --stats.
usage: pmc_block.py [iters] [fused 0|1]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import ops
from nemar_amd.models import networks

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ops.fused_blocks(bool(int(sys.argv[2])) if len(sys.argv) > 2 else True)
ops.side_stream(False)
dev = torch.device('cuda:0')
N, C, H, W = 16, 256, 64, 64
blocks = [networks.ResnetBlock(C, 'reflect', 'instance', True, True).to(dev) for _ in range(2)]
blocks[0].feeds_block = True
for b in blocks:
    for p in b.parameters():
        torch.nn.init.normal_(p, 0.0, 0.03)
ops.invalidate_packed_weights()
opt = ops.FlatAdam([p for b in blocks for p in b.parameters()])
x0 = torch.randn(N, C, H, W, device=dev)
g = torch.randn(N, C, H, W, device=dev)
for _ in range(iters):
    x = x0.clone().requires_grad_(True)
    h = ops.instance_norm(x, act=ops.ACT_RELU, planes=True)
    for b in blocks:
        h = b(h)
    torch.autograd.backward([h], [g])
    ops.join_side()
torch.cuda.synchronize()
