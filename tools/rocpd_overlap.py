"""Overlap report of the two-chain bench out of a rocprofv3 rocpd DB of `bench.py --steps 1 --warmup 1` (review r4 item 6).

Window = `evals` consecutive network evaluations of the TIMED sample (the level-0 FPS launches mark the evaluations, as in
rocpd_window.py; with two chains there are two such launches per evaluation). Reported per evaluation (32 patches):

 (i)   wall time during which no GEMM-shaped kernel (voxel convolution, 1x1 GEMM) is resident; during which none of ANY
       kind is; during which exactly one / two or more GEMM-shaped kernels are;
 (ii)  the critical path of one chain: walk back from the chain's last kernel of an evaluation, the predecessor of a kernel
       being the kernel OF THE SAME CHAIN (its main or its geometry queue) that ended last before it started -- a launch
       cannot start before everything it depends on has ended, and inside a captured graph it starts as soon as that has
       happened, so the latest-ending earlier kernel is the one that released it. Per kernel name: count on the path, time
       on the path, and the gaps (start - predecessor's end) charged to the released kernel;
 (iii) CU-time idle: 256 CUs x wall minus the integral of min(256, sum of resident workgroups / workgroups-per-CU bound 1)
       -- a LOWER bound of the idle CU time (a workgroup is counted as a whole CU for its kernel's whole life).
usage: rocpd_overlap.py <db> [evals=8] [tail=1]"""
import sqlite3
import sys
from collections import defaultdict

GEMM = ("conv3d_k3", "pw_split_kernel", "pw_wide_kernel", "pw_pp512_kernel", "pw_conv_kernel")


def is_gemm(name):
    return any(g in name for g in GEMM)


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n[:64]


def union_len(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def depth_hist(iv, t0, t1):
    """time spent at each overlap depth (0, 1, 2+) of the intervals inside [t0, t1)"""
    ev = []
    for s, e in iv:
        s, e = max(s, t0), min(e, t1)
        if e > s:
            ev.append((s, 1))
            ev.append((e, -1))
    ev.sort()
    hist = defaultdict(int)
    d, last = 0, t0
    for t, k in ev:
        hist[min(d, 2)] += t - last
        last = t
        d += k
    hist[min(d, 2)] += t1 - last
    return hist


def main(path, evals=8, tail=1):
    evals, tail = int(evals), int(tail)
    cur = sqlite3.connect(path).cursor()
    marker = "%fps_kernel<512, 16%"
    fps = list(cur.execute("select start, queue_id from kernels where name like ? order by start", (marker,)))
    qs = sorted({q for _, q in fps[-40:]})
    nch = len(qs)  # geometry queues = chains
    per_q = {q: [s for s, qq in fps if qq == q] for q in qs}
    # evaluation boundaries of chain 0: its FPS launches
    marks = per_q[qs[0]]
    i1 = len(marks) - tail - 1
    i0 = i1 - evals
    t0, t1 = marks[i0], marks[i1]
    rows = list(cur.execute("select name, queue_id, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) "
                            "from kernels where end>? and start<? order by start", (t0, t1)))
    wall = t1 - t0
    print(f"# {path.split('/')[-1]}: {evals} evaluations of the timed sample, {nch} chain(s), wall {wall / 1e3 / evals:.1f} us per evaluation "
          f"({sum(1 for r in rows if r[2] >= t0) / evals:.0f} launches per evaluation over all chains)")
    # ---- (i)
    allk = [(s, e) for _, _, s, e, _ in rows]
    gem = [(s, e) for n, _, s, e, _ in rows if is_gemm(n)]
    h_all = depth_hist(allk, t0, t1)
    h_gem = depth_hist(gem, t0, t1)
    gsum = sum(min(e, t1) - max(s, t0) for s, e in gem)
    print("## (i) residency, us per evaluation")
    print(f"no kernel of any kind resident: {h_all[0] / 1e3 / evals:.1f}")
    print(f"no GEMM-shaped kernel resident:  {h_gem[0] / 1e3 / evals:.1f}   ({h_gem[0] / wall * 100:.1f} % of the wall)")
    print(f"exactly one GEMM-shaped kernel:  {h_gem[1] / 1e3 / evals:.1f}")
    print(f"two or more GEMM-shaped kernels: {h_gem[2] / 1e3 / evals:.1f}")
    print(f"sum of GEMM-shaped kernel durations (as they ran, overlapped): {gsum / 1e3 / evals:.1f}")
    # ---- queues -> chains: a main queue belongs to the geometry queue whose evaluation marks it follows most closely
    queues = sorted({q for _, q, _, _, _ in rows})
    byq = defaultdict(list)
    for r in rows:
        byq[r[1]].append(r)
    print("## queues")
    for q in queues:
        k = byq[q]
        busy = sum(e - s for _, _, s, e, _ in k)
        g = sum(e - s for n, _, s, e, _ in k if is_gemm(n))
        print(f"queue {q}: {len(k) / evals:.1f} launches per evaluation, busy {busy / 1e3 / evals:.1f} us, of it GEMM-shaped {g / 1e3 / evals:.1f} us")
    mains = [q for q in queues if q not in qs]
    chain_of = {}
    for gq in qs:
        chain_of[gq] = gq
    # pair main queues with geometry queues by launch order at capture: the k-th main queue pairs with the k-th geometry queue
    # (checked below through the dependency heuristic: a main-queue pw_pp512 launch starts after ITS chain's marker)
    pp = {q: [s for n, _, s, e, _ in byq[q] if "pw_pp512_kernel<true, true>" in n] for q in mains}
    for mq in mains:
        best, bd = None, None
        for gq in qs:
            m = [x for x in per_q[gq] if t0 <= x < t1]
            # mean distance from each marker to the next pp512 launch on mq
            d = []
            for x in m:
                nxt = [p for p in pp[mq] if p > x]
                if nxt:
                    d.append(nxt[0] - x)
            if d:
                md = sum(d) / len(d)
                if bd is None or md < bd:
                    best, bd = gq, md
        chain_of[mq] = best if best is not None else qs[0]
    print("queue -> chain:", {q: chain_of[q] for q in queues})
    # ---- (ii) critical path of chain qs[0]
    for ch in qs[:1]:
        k = sorted([r for r in rows if chain_of.get(r[1]) == ch], key=lambda r: r[3])  # by end
        ends = [r[3] for r in k]
        m = [x for x in per_q[ch] if t0 <= x <= t1]
        on_path = defaultdict(lambda: [0, 0, 0])
        total_k = total_g = 0
        import bisect

        # walk back from the last kernel that ended before the window's end to the window's start
        j = bisect.bisect_right(ends, t1) - 1
        curk = k[j]
        path_rows = []
        while curk[2] > t0:
            # predecessor: latest end <= cur.start (+ 1 us of slack for timestamp jitter)
            p = bisect.bisect_right(ends, curk[2] + 1000) - 1
            while p >= 0 and (k[p] is curk or k[p][2] >= curk[2]):
                p -= 1
            if p < 0:
                break
            gap = max(0, curk[2] - k[p][3])
            st = on_path[short(curk[0])]
            st[0] += 1
            st[1] += curk[3] - curk[2]
            st[2] += gap
            total_k += curk[3] - curk[2]
            total_g += gap
            path_rows.append((curk[1], curk[2], curk[3], gap, short(curk[0])))
            curk = k[p]
        print(f"## (ii) critical path of the chain on geometry queue {ch}: us per evaluation; kernels {total_k / 1e3 / evals:.1f} + gaps {total_g / 1e3 / evals:.1f} "
              f"= {(total_k + total_g) / 1e3 / evals:.1f} of {wall / 1e3 / evals:.1f} wall; {len(path_rows) / evals:.1f} launches on the path")
        print("on_path_per_eval,kernel_us_per_eval,gap_before_us_per_eval,kernel")
        for n, (c, d, g) in sorted(on_path.items(), key=lambda x: -(x[1][1] + x[1][2])):
            print(f"{c / evals:.1f},{d / 1e3 / evals:.1f},{g / 1e3 / evals:.1f},{n}")
        geo_on = sum(1 for r in path_rows if r[0] == ch)
        print(f"# launches of the path that ran on the geometry queue: {geo_on / evals:.1f} per evaluation")
    # ---- (iii)
    ev = []
    for n, q, s, e, wgs in rows:
        s, e = max(s, t0), min(e, t1)
        if e > s:
            c = min(256, max(1, int(wgs)))
            ev.append((s, c))
            ev.append((e, -c))
    ev.sort()
    used, d, last = 0, 0, t0
    for t, c in ev:
        used += min(256, d) * (t - last)
        last = t
        d += c
    tot = 256 * wall
    print(f"## (iii) CU-time: {(tot - used) / tot * 100:.1f} % idle at least (every resident workgroup counted as one whole CU, capped at 256)")


if __name__ == "__main__":
    main(*sys.argv[1:])
