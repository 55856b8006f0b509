"""Simulation for a block-pruned exact FPS on large clouds: points Morton-sorted, fixed blocks of BS consecutive points with tight
boxes (instead of the 16^3 grid's cells, which hold ~100 points apiece on surface clouds and 0 on most of the grid), blocks dealt
to the 16 waves round-robin. Per round: which blocks can change (box lower bound < block maximum), how they spread over the waves,
how many sequential L2 round trips the slowest wave needs when it takes 64 / BS hit blocks per pass. Statistics only (float64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from oracle import net_ref


def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-30) * (2 ** bits - 1)).astype(np.uint64)
    code = np.zeros(len(p), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code


def run(p, m, bs, waves=16, bits=10):
    n = len(p)
    perm = np.random.default_rng(1).permutation(n)  # (arbitrary order inside a bin: the build scatters with atomics)
    order = perm[np.argsort(morton(p[perm], bits), kind="stable")]
    ps = p[order]
    nb = (n + bs - 1) // bs
    pad = nb * bs - n
    pp = np.concatenate([ps, np.repeat(ps[-1:], pad, 0)]) if pad else ps
    blk = pp.reshape(nb, bs, 3)
    lo, hi = blk.min(1), blk.max(1)
    dist = np.full(nb * bs, 1e38)
    if pad:
        dist[n:] = -1.0
    dist = dist.reshape(nb, bs)
    bmax = dist.max(1)
    wave = np.arange(nb) % waves
    per_pass = max(1, 64 // bs)
    cur = ps[np.where(order == 0)[0][0]]
    hits_tot = passes_tot = maxw_tot = upd_tot = 0
    hist = np.zeros(12, dtype=np.int64)
    for j in range(1, m):
        d = np.maximum(np.maximum(lo - cur, cur - hi), 0.0)
        lb = (d * d).sum(1)
        hit = np.flatnonzero(lb < bmax)
        if j > 1:
            hits_tot += len(hit)
            hw = np.bincount(wave[hit], minlength=waves)
            mw = hw.max()
            maxw_tot += mw
            passes = -(-mw // per_pass)
            passes_tot += passes
            hist[min(passes, 11)] += 1
        dd = ((blk[hit] - cur) ** 2).sum(2)
        nd = np.minimum(dist[hit], dd)
        upd_tot += int((nd != dist[hit]).sum()) if j > 1 else 0
        dist[hit] = nd
        bmax[hit] = nd.max(1)
        b = int(bmax.argmax())
        k = int(dist[b].argmax())
        cur = blk[b, k]
    r = m - 2
    print(f"BS={bs:3d}: blocks {nb:5d} ({nb / (waves * 64):.2f} per lane)  hit blocks/round {hits_tot / r:6.2f}  points loaded/round {hits_tot * bs / r:7.1f}"
          f"  changed/round {upd_tot / r:6.1f}  max hits on a wave {maxw_tot / r:5.2f}  passes/round {passes_tot / r:5.2f}  hist(passes) {hist.tolist()}")


if __name__ == "__main__":
    n, m = 50000, 12500
    for kind in ("patches", "volume"):
        if kind == "patches":
            p = net_ref.synthetic_patches(1, n, seed=0)[0][0].numpy().T.astype(np.float64)
        else:
            p = np.random.default_rng(0).random((n, 3))
        print(kind)
        for bits in (5, 6):
            print(" bits per axis", bits)
            for bs in (16, 32):
                run(p, m, bs, bits=bits)
