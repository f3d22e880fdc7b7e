"""Error of a Winograd F(2x2, 3x3) form of the wide 3x3 layers under the fp16 x 3 operand split, against the direct form.
numpy model of the arithmetic: fp32 transforms, operands v 2^k = h + l (fp16 each), products h h' + h l' + l h' accumulated in fp32 in
16-element groups (the 16-bit MFMA adds 16 products before it rounds), float64 reference.  256 input channels, 64 outputs, 32 x 32 map,
values distributed like the residual blocks' (InstanceNorm + ReLU activations, N(0, 0.02)-scale weights after some training drift).
  python tools/winograd_error.py"""
import numpy as np

rng = np.random.default_rng(0)
C, K, H = 256, 64, 32
x = np.maximum(rng.standard_normal((C, H + 2, H + 2)), 0).astype(np.float32)          # padded input
w = (rng.standard_normal((K, C, 3, 3)) * 0.05).astype(np.float32)


def split(v):
    """-> (h, l) fp16 planes of v * 2^k (k: max |v 2^k| in [2^11, 2^12)) as float32 values, and 2^-k"""
    m = float(np.abs(v).max())
    k = 11 - int(np.floor(np.log2(m))) if m > 0 else 0
    s = np.float32(2.0 ** k)
    vs = v * s
    h = vs.astype(np.float16).astype(np.float32)
    l = (vs - h).astype(np.float16).astype(np.float32)
    return h, l, np.float32(2.0 ** -k)


def dot3(a, b, axis_groups=16):
    """sum over the last axis of a (.., R) against b (.., R): three partial products, fp32 accumulation of exact 16-product groups"""
    ah, al, sa = split(a)
    bh, bl, sb = split(b)
    R = a.shape[-1]
    acc = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), dtype=np.float32)
    for lo in range(0, R, axis_groups):
        sl = slice(lo, lo + axis_groups)
        for p, q in ((al, bh), (ah, bl), (ah, bh)):
            acc = (acc + (p[..., sl].astype(np.float64) * q[..., sl].astype(np.float64)).sum(-1).astype(np.float32)).astype(np.float32)
    return acc * (sa * sb)


# ---- float64 reference and the direct fp16 x 3 form (reduction order: channels inside a tap, taps outside, as igemm_split16_kernel) ----
ref = np.zeros((K, H, H))
for r in range(3):
    for s in range(3):
        ref += np.einsum('kc,chw->khw', w[:, :, r, s].astype(np.float64), x[:, r:r + H, s:s + H].astype(np.float64))
cols = np.stack([x[:, r:r + H, s:s + H] for r in range(3) for s in range(3)], 0)       # [9, C, H, W]
A = w.transpose(0, 2, 3, 1).reshape(K, 9 * C)                                           # [K, 9 C]
Bm = cols.reshape(9 * C, H * H).T                                                       # [HW, 9 C]
direct = dot3(A[:, None, :], Bm[None, :, :]).reshape(K, H, H)

# ---- Winograd F(2x2, 3x3): U = G g G^T (fp32), V = B^T d B (fp32), M = sum_c U V per transform position (fp16 x 3), Y = A^T M A (fp32) ----
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float32)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float32)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float32)
U = np.einsum('ij,kcjl,ml->kcim', G, w, G).astype(np.float32)                           # [K, C, 4, 4]
T = H // 2
d = np.stack([np.stack([x[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for tx in range(T)], 1) for ty in range(T)], 1)    # [C, T, T, 4, 4]
V = np.einsum('ij,cyxjl,ml->cyxim', Bt, d, Bt).astype(np.float32)
M = np.zeros((K, T, T, 4, 4), dtype=np.float32)
for i in range(4):
    for j in range(4):
        M[:, :, :, i, j] = dot3(U[:, None, :, i, j], V[:, :, :, i, j].reshape(C, T * T).T[None, :, :]).reshape(K, T, T)
Y = np.einsum('ij,kyxjl,ml->kyxim', At, M, At).astype(np.float32)                        # [K, T, T, 2, 2]
wino = Y.transpose(0, 1, 3, 2, 4).reshape(K, H, H)

# ---- the same two forms in plain fp32 (what the split has to match) ----
fp32 = np.zeros((K, H, H), dtype=np.float32)
for r in range(3):
    for s in range(3):
        fp32 += np.einsum('kc,chw->khw', w[:, :, r, s], x[:, r:r + H, s:s + H]).astype(np.float32)
scale = np.abs(ref).max()
for name, y in (('direct, fp16 x 3', direct), ('Winograd F(2x2,3x3), fp16 x 3 on the transformed tiles', wino), ('direct, plain fp32 (numpy einsum)', fp32)):
    e = np.abs(y - ref)
    print('%-56s max |err| %.3e   rms %.3e   (relative to max |y| = %.2f: %.2e / %.2e)' % (name, e.max(), np.sqrt((e ** 2).mean()), scale, e.max() / scale,
                                                                                           np.sqrt((e ** 2).mean()) / scale))
print('multiplies per 2x2 outputs and (k, c): direct 36, Winograd 16 (2.25x fewer MFMAs); transformed input = 4x the elements of the input')
