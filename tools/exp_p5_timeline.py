"""Phases of a workgroup of pw_pp512_kernel (512 -> 1024, P = 8192, B = 32 as the sampler launches it) from a -DPP_TIMELINE
build of pointwise.hip (tools/build_pw_variant.sh tl "-DPP_TIMELINE"; P2PB_LIB_PATH=tools/exp/lib_pwtl.so, P2PB_EXPERIMENT="pw_pp=1"):
s_memtime stamps of wave 0 (half 0: multiply, then stage) and wave 4 (half 1: stage, then multiply), stored last."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import _lib
buf = torch.zeros(32 * 8192, dtype=torch.int64, device="cuda")
assert _lib.lib().p2pb_pp_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_pw.py"))
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 32).astype(np.float64)
t = t[t[:, 0] != 0]
q = lambda a: f"{np.mean(a):.0f} (p10 {np.percentile(a, 10):.0f}, p90 {np.percentile(a, 90):.0f})"
print(f"{len(t)} workgroups ({len(t) / 256:.0f} in sequence per CU); shader cycles")
print("life", q(t[:, 3] - t[:, 0]), "| prologue", q(t[:, 1] - t[:, 0]), "| stage loop", q(t[:, 2] - t[:, 1]), f"-> per stage {np.mean(t[:, 2] - t[:, 1]) / 16:.0f} (matrix pipe alone: 3072)",
      "| epilogue", q(t[:, 3] - t[:, 2]))
a, c = t[:, 4:12], t[:, 20:28]
for name, w in (("wave 0", a), ("wave 4", c)):
    print(f"stage 6, {name}: scalars + take + loads + fragments + 48 MFMAs with the window interleaved", q(w[:, 5] - w[:, 0]),
          "| wait everything landed", q(w[:, 6] - w[:, 5]), "| barrier", q(w[:, 7] - w[:, 6]), "| total", q(w[:, 7] - w[:, 0]))
