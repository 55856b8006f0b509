"""CPU simulation (numpy, float64) of an EXACT multi-sample FPS round for the pruned large-cloud kernel (round 6):
every wave of fps_grid_kernel offers its best cell maximum w and an upper bound s on everything else it holds; the 16 offers sorted by
key, offer i is sample i of the round iff its cell is not reached by a better offer's point (the kernel's own box test) and s_j < w_i
for every better offer j. Prints the samples per round and checks the emitted sequence against plain FPS.
    python tools/exp/fps/multi_emit_sim.py patch | room | volume      (50000 -> 12500; results: profiles/r06_fps_multi_emit.txt)"""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from oracle import net_ref
kind = sys.argv[1] if len(sys.argv) > 1 else "patch"
N, M, G = 50000, 12500, 16
rng = np.random.default_rng(0)
if kind == "patch":
    x = net_ref.synthetic_patches(1, N, seed=0)[0][0].numpy().T.astype(np.float64)
elif kind == "room":
    x = rng.random((N, 3)) * 2 - 1; ax = rng.integers(0, 3, N); x[np.arange(N), ax] = rng.integers(0, 2, N) * 2 - 1.0
else:
    x = rng.random((N, 3)) * 2 - 1
lo, hi = x.min(0), x.max(0)
cell = np.minimum(((x - lo) / np.maximum(hi - lo, 1e-12) * G).astype(int), G - 1)
cid = (cell[:, 2] * G + cell[:, 1]) * G + cell[:, 0]
occ = np.unique(cid)
cells = [np.flatnonzero(cid == c) for c in occ]
wave_of = np.array([((c % 16) + 3 * ((c // 16) % 16) + 9 * (c // 256)) % 16 for c in occ])
blo = np.array([x[idx].min(0) for idx in cells]); bhi = np.array([x[idx].max(0) for idx in cells])
d = np.full(N, 1e38); d = np.minimum(d, ((x - x[0]) ** 2).sum(1))
seq = [0]; hist = {}
def cellstats():
    m1 = np.array([d[idx].max() for idx in cells]); am = np.array([idx[d[idx].argmax()] for idx in cells])
    m2 = np.array([np.partition(d[idx], -2)[-2] if len(idx) > 1 else -1.0 for idx in cells])
    return m1, m2, am
t0 = time.time()
while len(seq) < M:
    m1, m2, am = cellstats()
    # per wave: best cell (w), runner-up bound s = max(second best cell max in the wave, m2 of the best cell)
    W = []
    for w in range(16):
        ids = np.flatnonzero(wave_of == w)
        if len(ids) == 0: continue
        o = ids[np.argsort(-m1[ids])]
        b = o[0]; s = max(m2[b], m1[o[1]] if len(o) > 1 else -1.0)
        W.append((m1[b], s, b))
    W.sort(key=lambda t: -t[0])
    emitted = []
    for i, (wv, s, b) in enumerate(W):
        ok = True
        for j in range(i):
            e = x[am[W[j][2]]]
            bd = np.maximum(np.maximum(blo[b] - e, e - bhi[b]), 0.0)
            if not ((bd ** 2).sum() >= m1[b]) or not (W[j][1] < wv): ok = False; break
        if not ok: break
        emitted.append(am[b])
    hist[len(emitted)] = hist.get(len(emitted), 0) + 1
    for e in emitted:
        d = np.minimum(d, ((x - x[e]) ** 2).sum(1)); seq.append(int(e))
print(kind, "rounds", sum(hist.values()), "samples", len(seq), "avg per round %.2f" % (len(seq) / sum(hist.values())), dict(sorted(hist.items())), "%.1fs" % (time.time() - t0))
dd = np.full(N, 1e38); cur = 0; ref = [0]
for jj in range(1, len(seq)):
    dd = np.minimum(dd, ((x - x[cur]) ** 2).sum(1)); cur = int(dd.argmax()); ref.append(cur)
print("sequence equals plain FPS:", ref == seq)
