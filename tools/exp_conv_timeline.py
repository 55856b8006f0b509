"""Where a workgroup of the pre-split brick-list convolution (fp_layers.3.1.voxel_layers.0, C64 -> 64, r = 32, the
launch of one bench evaluation) spends its life: a -DCONV_TIMELINE build (tools/build_conv_variants.sh tl:"-DCONV_TIMELINE",
P2PB_LIB_PATH=tools/exp/lib_tl.so) stamps s_memtime at start / first DMA issued / every stage release and end of taps /
accumulators final / epilogue done in wave 0 of every workgroup, with its XCC / HW id; this prints the mean phase lengths,
and for the workgroups that share a CU how their matrix phases overlap."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WHICH", "brick")
from p2p_bridge_amd import _lib  # noqa: E402

lib = _lib.lib()
buf = torch.zeros(16 * 65536, dtype=torch.int64, device="cuda")
rc = lib.p2pb_conv_timeline_set(ctypes.c_void_p(buf.data_ptr()))
assert rc == 0, rc
import builtins  # noqa: E402
import runpy  # noqa: E402

builtins._p2pb_tl_buf = buf

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_conv_instances.py"))  # issues the launch 4 x
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 16)
t = t[t[:, 1] != 0]
ids, st = t[:, 0], t[:, 1:].astype(np.float64)
q = lambda a: f"{np.mean(a):.0f} (p10 {np.percentile(a, 10):.0f}, p90 {np.percentile(a, 90):.0f})"
if os.environ["WHICH"] == "compact":
    # slots: 0 start, 1 list in LDS, 2 first DMA issued, 3 + 2k / 4 + 2k stage k released / taps issued (k < 4), 11 loop done,
    # 12 class constants staged, 13 active outputs stored, 14 constants + statistics written
    print(f"{len(t)} workgroups (compact form); shader cycles (s_memtime), wave 0")
    print("life", q(st[:, 14] - st[:, 0]))
    print("start -> brick list in LDS", q(st[:, 1] - st[:, 0]))
    has = st[:, 2] != 0
    print(f"workgroups with a stage loop: {has.sum()}")
    s2 = st[has]
    print("list -> first DMA issued", q(s2[:, 2] - s2[:, 1]))
    print("first DMA issued -> stage 0 released", q(s2[:, 3] - s2[:, 2]))
    for k in range(4):
        print(f"stage {k}: taps", q(s2[:, 4 + 2 * k] - s2[:, 3 + 2 * k]), "; to the next release", q((s2[:, 5 + 2 * k] if k < 3 else s2[:, 4 + 2 * k]) - s2[:, 4 + 2 * k]))
    print("stage loop in all (first release -> loop done)", q(s2[:, 11] - s2[:, 3]))
    print("class constants through LDS", q(s2[:, 12] - s2[:, 11]))
    print("epilogue of the active outputs (-> barrier)", q(s2[:, 13] - s2[:, 12]))
    print("constants + statistics of the other voxels", q(s2[:, 14] - s2[:, 13]))
    sys.exit(0)
nst = int(((st[0, 2:12] != 0).sum()) // 2)
print(f"{len(t)} workgroups, {nst} stages; all times in shader cycles (s_memtime), wave 0 of each workgroup")
print("life", q(st[:, 14] - st[:, 0]), "; start -> first DMA issued", q(st[:, 1] - st[:, 0]), "; -> stage 0 released", q(st[:, 2] - st[:, 1]))
for k in range(nst):
    rel, end = st[:, 2 + 2 * k], st[:, 3 + 2 * k]
    nxt = st[:, 4 + 2 * k] if k + 1 < nst else st[:, 13]
    print(f"stage {k}: taps", q(end - rel), "; end of taps -> next release / accumulators final", q(nxt - end))
print("epilogue (accumulators final -> stores acknowledged)", q(st[:, 14] - st[:, 13]))
hw = ids & 0xffffffff
xcc = (ids >> 32) & 0xf
cu = (xcc << 16) | (hw & 0xff00)
ev = []
for c in np.unique(cu)[:64]:
    rows = st[cu == c]
    pts = []
    for r in rows:
        for k in range(nst):
            pts.append((r[2 + 2 * k], 1)); pts.append((r[3 + 2 * k], -1))
    pts.sort()
    cur, last, one, two = 0, pts[0][0], 0.0, 0.0
    for x, d in pts:
        if cur >= 1: one += x - last
        if cur >= 2: two += x - last
        cur += d; last = x
    tot = rows[:, 14].max() - rows[:, 0].min()
    ev.append((one / tot, two / tot, len(rows)))
ev = np.array(ev)
print(f"{len(np.unique(cu))} distinct (xcc, cu) ids; per CU (first 64): workgroups {ev[:, 2].mean():.1f}; time with >= 1 workgroup in a taps phase {ev[:, 0].mean():.2f}, with >= 2 {ev[:, 1].mean():.2f} of the CU's span")
