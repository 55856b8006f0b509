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
import runpy  # noqa: E402

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_conv_instances.py"))  # issues the launch 4 x
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 16)
t = t[t[:, 1] != 0]
ids, st = t[:, 0], t[:, 1:].astype(np.float64)
nst = int(((st[0, 2:13] != 0).sum()) // 2)
print(f"{len(t)} workgroups, {nst} stages; all times in shader cycles (s_memtime), wave 0 of each workgroup")
life = st[:, 14] - st[:, 0]
print(f"life {life.mean():.0f} (min {life.min():.0f}, max {life.max():.0f}); prologue to first DMA issued {np.mean(st[:, 1] - st[:, 0]):.0f}; "
      f"first DMA issued -> stage 0 released {np.mean(st[:, 2] - st[:, 1]):.0f}")
for k in range(nst):
    rel, end = st[:, 2 + 2 * k], st[:, 3 + 2 * k]
    nxt = st[:, 4 + 2 * k] if k + 1 < nst else st[:, 13]
    print(f"stage {k}: taps {np.mean(end - rel):.0f} (min {np.min(end - rel):.0f}); end of taps -> next release / accumulators final {np.mean(nxt - end):.0f}")
print(f"epilogue (accumulators final -> stores acknowledged) {np.mean(st[:, 14] - st[:, 13]):.0f}")
# co-residency: same XCC and same CU (HW_ID: cu_id bits 11:8, sh 12, se 15:13 on gfx9)
hw = ids & 0xffffffff
xcc = (ids >> 32) & 0xf
cu = (xcc << 16) | (hw & 0xff00)
order = np.argsort(st[:, 0])
span = st[:, 14].max() - st[:, 0].min()
print(f"launch span {span:.0f} cycles; {len(np.unique(cu))} distinct (xcc, cu) ids")
# matrix-phase overlap on each CU: fraction of the time at least one / both resident workgroups are inside a taps phase
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
print(f"per CU (first 64): workgroups {ev[:, 2].mean():.1f}; time with >= 1 workgroup in a taps phase {ev[:, 0].mean():.2f}, with >= 2 {ev[:, 1].mean():.2f} of the CU's span")
