"""Time every traced conv call shape (tools/trace_convs.py output) and rank by time x count per step."""
import ctypes, json, os, sys
os.environ.setdefault('NEMAR_AB_LIBRARY', '1')      # nemar_tune*: the measurement build of the library (nemar_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemar_amd import _lib
from tests.side_inputs import SideInputs
from tools.microbench import timeit

lib = SideInputs(_lib.load()); dev = torch.device('cuda:0')
for kv in sys.argv[2:]:
    k, v = kv.split('='); lib.tune(int(k), int(v))
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
rows = []
ROUTES = {0: "exact", 1: "narrow", 2: "split16", 3: "s16g", 4: "k7"}
arena = None
for line in open(sys.argv[1]):
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    op, C0, C1, K, R, s, p, pm, H, W, N = (d[k] for k in ("op", "C0", "C1", "K", "R", "stride", "pad", "pad_mode", "H", "W", "N"))
    act = d.get("act", 1)
    C = C0 + C1
    OH, OW = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    x0 = torch.randn(N, C0, H, W, device=dev); x1 = torch.randn(N, C1, H, W, device=dev) if C1 else None
    w = torch.randn(K, C, R, R, device=dev) * 0.05; b = torch.randn(K, device=dev)
    y = torch.empty(N, K, OH, OW, device=dev); gy = torch.randn(N, K, OH, OW, device=dev)
    gx0 = torch.empty(N, C0, H, W, device=dev); gx1 = torch.empty(N, C1, H, W, device=dev) if C1 else None
    gw = torch.zeros_like(w)
    need = lib.conv2d_scratch(N, H, W, K, C, R, R, s, p)       # the arena ops.py registers: wide layers on the split-16 kernels
    if need and (arena is None or arena.numel() * 4 < need):
        arena = torch.empty(need // 4 + 16, device=dev)
    if arena is not None:
        lib.set_scratch(P(arena), arena.numel() * 4)
    wsb = max(lib.conv2d_fwd_workspace(N, H, W, K, C, R, R, s, p), lib.conv2d_bwd_data_workspace(N, C, H, W, K, R, R, s, p, pm))
    ws = torch.empty(wsb // 4 + 16, device=dev)
    if op == "fwd":
        f = lambda pre: lib.conv2d_fwd(P(x0), C0, P(x1), C1, P(w), P(b), P(y), N, H, W, K, R, R, s, p, pm, act, 0.2, P(ws), wsb, pre, st())
    elif op == "dgrad":
        f = lambda pre: lib.conv2d_bwd_data(P(gy), P(w), None, 0, 0.0, P(gx0), C0, P(gx1), C1, N, H, W, K, OH, OW, R, R, s, p, pm, P(ws), wsb, pre, st())
    else:
        wwb = lib.conv2d_bwd_weight_workspace(N, C, H, W, K, OH, OW, R, R, s, p)
        ws3 = torch.empty(wwb // 4 + 16, device=dev)
        f = lambda pre, ws3=ws3, wwb=wwb: lib.conv2d_bwd_weight(P(x0), C0, P(x1), C1, P(gy), P(gw), P(b), N, H, W, K, OH, OW, R, R, s, p, pm, P(ws3), wwb, st())
    if op == "fwd" and arena is not None and act == 0:
        pass
    f(0)
    d["route"] = ROUTES.get(lib.last_route(), "?")
    t = min(timeit(lambda: f(1), 10, 2) for _ in range(3))      # min of three: an allocator hiccup once showed up as a 3.9 ms 'layer'
    flop = 2.0 * N * K * OH * OW * C * R * R
    rows.append((t * d["count"], t, flop, d))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print("total conv time per step: %.2f ms" % (tot * 1e3))
for tt, t, flop, d in rows:
    print("%6.2f ms  x%-3d %8.1f us %6.1f TF  %-7s %-5s N=%d C=%d+%d K=%d k%d s%d p%d pm%d %dx%d" % (
        tt * 1e3, d["count"], t * 1e6, flop / t / 1e12, d["route"], d["op"], d["N"], d["C0"], d["C1"], d["K"], d["R"], d["stride"], d["pad"], d["pad_mode"], d["H"], d["W"]))
