"""Phases of a workgroup of the ping-pong GEMM (512 -> 1024, P = 8192, B = 32 as the sampler launches it): a -DPP_TIMELINE
build of pointwise.hip (tools/build_pw_variant.sh tl "-DPP_TIMELINE"; P2PB_LIB_PATH=tools/exp/lib_pwtl.so) keeps s_memtime
at kernel start / after the prologue's barrier / after the stage loop / inside and after the epilogue in wave 0 and stores them last."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p2p_bridge_amd import _lib
buf = torch.zeros(16 * 8192, dtype=torch.int64, device="cuda")
assert _lib.lib().p2pb_pp_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_pw.py"))
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
t = t[t[:, 0] != 0]
q = lambda a: f"{np.mean(a):.0f} (p10 {np.percentile(a, 10):.0f}, p90 {np.percentile(a, 90):.0f})"
print(f"{len(t)} workgroups (one per CU at a time, {len(t) / 256:.0f} in sequence per CU); shader cycles")
print("life", q(t[:, 3] - t[:, 0]))
print("prologue (start -> first stage released)", q(t[:, 1] - t[:, 0]))
print("stage loop, 16 stages", q(t[:, 2] - t[:, 1]), "-> per stage", f"{np.mean(t[:, 2] - t[:, 1]) / 16:.0f}", "(matrix pipe alone: 3072)")
print("epilogue (statistics, extrema, stores acknowledged)", q(t[:, 3] - t[:, 2]))
print("  bias table + scale", q(t[:, 4] - t[:, 2]), "| optional store + statistics", q(t[:, 5] - t[:, 4]), "| extrema + stores acknowledged", q(t[:, 3] - t[:, 5]))
a, c = t[:, 6:11], t[:, 11:16]
print("inside the even stage 6 -- half 0 (wave 0): DMA issue + multiply", q(a[:, 1] - a[:, 0]), "| wait for the raw activations", q(a[:, 2] - a[:, 1]),
      "| stage B + loads", q(a[:, 3] - a[:, 2]), "| barrier", q(a[:, 4] - a[:, 3]))
print("                           half 1 (wave 4): multiply", q(c[:, 1] - c[:, 0]), "| barrier", q(c[:, 2] - c[:, 1]),
      "| DMA issue + wait + stage B + loads", q(c[:, 3] - c[:, 2]))
print("   half 1's interval starts", q(c[:, 0] - a[:, 0]), "cycles after half 0's; its barrier releases", q(c[:, 2] - a[:, 4]), "after half 0 leaves it")
span = t[:, 3].max() - t[:, 0].min()
print(f"launch span {span:.0f} cycles; sum of lives / (256 CUs x span) = {np.sum(t[:, 3] - t[:, 0]) / (256 * span):.2f}")
