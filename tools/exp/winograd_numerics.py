"""Numerical price of Winograd F(2x2x2, 3x3x3) for the voxel convolutions under the f16x3 operand contract (CPU, numpy): a round-6
planning number, not a kernel. One 8^3 layer, Cin = Cout = C: direct correlation vs the Winograd form, both with operands cut to
22 significand bits (the fp16 pair) and fp32 accumulation over the input channels, against fp64.
    python tools/exp/winograd_numerics.py [C]"""
import sys

import numpy as np

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = 8
rng = np.random.default_rng(0)


def cut22(x):  # the fp16-pair representation of the scaled operand: 22 significand bits (round to nearest)
    x = np.asarray(x, np.float32)
    m, e = np.frexp(x)
    return np.ldexp(np.round(m.astype(np.float64) * 2.0 ** 22) / 2.0 ** 22, e).astype(np.float32)


x = rng.standard_normal((C, R, R, R)).astype(np.float32)
x = (x / (1 + np.exp(-x))).astype(np.float32)  # Swish of a normalised tensor
w = (rng.standard_normal((C, C, 3, 3, 3)) / np.sqrt(27 * C)).astype(np.float32)
xp = np.zeros((C, R + 2, R + 2, R + 2), np.float32)
xp[:, 1:-1, 1:-1, 1:-1] = x

# fp64 reference and the direct form in the contract's arithmetic
ref = np.zeros((C, R, R, R))
dirv = np.zeros((C, R, R, R), np.float32)
xq, wq = cut22(xp), cut22(w)
for a in range(3):
    for b in range(3):
        for c in range(3):
            sl = xp[:, a:a + R, b:b + R, c:c + R]
            ref += np.einsum("oi,idhw->odhw", w[:, :, a, b, c].astype(np.float64), sl.astype(np.float64))
            dirv += np.einsum("oi,idhw->odhw", wq[:, :, a, b, c], xq[:, a:a + R, b:b + R, c:c + R]).astype(np.float32)

# Winograd F(2,3) per axis: BT (4x4) on the input tile, G (4x3) on the kernel, AT (2x4) on the products
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)


def t3(M, t):  # apply M along the last three axes (fp32 arithmetic, as a kernel's VALU would)
    t = np.einsum("ad,...dhw->...ahw", M, t).astype(np.float32)
    t = np.einsum("bh,...ahw->...abw", M, t).astype(np.float32)
    return np.einsum("cw,...abw->...abc", M, t).astype(np.float32)


U = cut22(t3(G, w))  # [Co, Ci, 4, 4, 4]   (the weight pack: transformed once, then split)
T = R // 2
tiles = np.stack([xp[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4, 2 * k:2 * k + 4] for i in range(T) for j in range(T) for k in range(T)], 1)
V = cut22(t3(BT, tiles))  # [Ci, tiles, 4, 4, 4]  (the operand: transformed in fp32, then split)
Mx = np.einsum("oiabc,itabc->otabc", U, V).astype(np.float32)  # 64 GEMMs over the input channels, fp32 accumulate
Y = t3(AT, Mx)  # [Co, tiles, 2, 2, 2]
win = np.zeros((C, R, R, R), np.float32)
n = 0
for i in range(T):
    for j in range(T):
        for k in range(T):
            win[:, 2 * i:2 * i + 2, 2 * j:2 * j + 2, 2 * k:2 * k + 2] = Y[:, n]
            n += 1
scale = np.abs(ref).max()
for name, v in (("direct, 22-bit operands, fp32 accumulate", dirv), ("Winograd F(2,3)^3, same contract       ", win)):
    e = np.abs(v - ref)
    print(f"C = {C}: {name}: max |err| / max |y| = {e.max() / scale:.2e}, rms err / rms y = {np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")
print(f"products per output voxel and channel pair: direct 27, Winograd {64 / 8:.0f}")
