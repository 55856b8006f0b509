"""The GroupNorm that follows a 1x1 convolution, handed to the convolution's own entry point (p2pb_gn_finisher_arm, csrc/common.h
GnFinish): scale / shift / channel mean must be the SAME BITS as the separate p2pb_gn_affine_params launch gives on the same
partials -- one device function runs in every form: the launch behind the producer (default) and, under P2PB_GN_FINISH=7, the
producing kernel's last workgroup per (sample, group) -- with and without AdaGN styles, ragged channel tiles, repeated launches
(the ticket counters return to zero), two streams side by side. The whole file is re-run under P2PB_GN_FINISH=7 by its last test."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (B, Cin, Cout, P, groups): pw_split 128- and 256-channel tiles, groups narrower and wider than a tile, the ping-pong
    (4, 64, 64, 2048, 8),     # and wide forms (finished by the launch behind them), a ragged last channel tile
    (16, 64, 128, 8192, 8),
    (8, 128, 256, 2048, 8),
    (8, 256, 512, 8192, 8),   # ping-pong at b >= 8
    (2, 256, 512, 2048, 8),   # split, 256-channel tiles? (grid decides)
    (3, 96, 192, 1024, 8),    # 192 = 128 + 64: ragged tile, cg = 24 does not nest -> launch behind
    (4, 128, 1024, 512, 8),   # cg = 128
    (2, 32, 32, 4096, 8),     # narrow: wide kernel
    (5, 64, 256, 640, 4),     # cg = 64
]


def _run(b, ci, co, p, groups, styled, want_mean, seed):
    from p2p_bridge_amd import fused

    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(b, ci, p, device="cuda", generator=g)
    conv = torch.nn.Conv1d(ci, co, 1).cuda()
    gamma = torch.randn(co, device="cuda", generator=g)
    beta = torch.randn(co, device="cuda", generator=g)
    wide = torch.randn(b, 2 * co + 40, device="cuda", generator=g)
    style = wide[:, 8:8 + 2 * co] if styled else None  # (a column slice of a wider matrix: row stride passed through)
    fin = (float(p), groups, gamma, beta, style, 1e-5, want_mean)
    with torch.no_grad():
        y, st, (sc, sh, mean) = fused.pw_conv(x, conv, fin=fin)
        y2, st2 = fused.pw_conv(x, conv)
        ref = fused.gn_affine_params(st2, float(p), groups, gamma, beta, style, 1e-5, want_mean)
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(st, st2)
    assert torch.equal(sc, ref[0]) and torch.equal(sh, ref[1]), (b, ci, co, p, groups, (sc - ref[0]).abs().max().item())
    assert (mean is None) == (ref[2] is None) and (mean is None or torch.equal(mean, ref[2]))
    return sc


def _counts():
    import ctypes

    from p2p_bridge_amd._lib import lib

    a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    lib().p2pb_debug_gn_finisher(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


@pytest.mark.parametrize("b,ci,co,p,groups", SHAPES)
def test_finished_by_the_producer_equals_the_separate_launch(b, ci, co, p, groups):
    f0, b0 = _counts()
    for rep in range(3):  # (the counters must be back at zero: a second and third launch reuse them)
        _run(b, ci, co, p, groups, styled=rep != 1, want_mean=rep == 2, seed=rep)
    f1, b1 = _counts()
    assert (f1 - f0) + (b1 - b0) == 3
    print(f"\n({b}, {ci}, {co}, {p}, groups {groups}): inside the producer {f1 - f0}, launch behind {b1 - b0}", end="")
    if os.environ.get("P2PB_GN_FINISH") == "7" and (b, ci, co, p) != (3, 96, 192, 1024):  # (cg = 24 does not nest in a tile)
        assert f1 - f0 == 3
    if os.environ.get("P2PB_GN_FINISH", "0") == "0":
        assert b1 - b0 == 3


def test_two_streams_side_by_side():
    from p2p_bridge_amd import fused

    torch.manual_seed(0)
    convs = [torch.nn.Conv1d(64, 128, 1).cuda() for _ in range(2)]
    xs = [torch.randn(16, 64, 8192, device="cuda") for _ in range(2)]
    gam, bet = torch.randn(128, device="cuda"), torch.randn(128, device="cuda")
    fin = (8192.0, 8, gam, bet, None, 1e-5, False)
    with torch.no_grad():
        ref = [fused.pw_conv(x, c, fin=fin)[2] for x, c in zip(xs, convs)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rep in range(20):
            outs = []
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    for _ in range(4):
                        outs.append((k, fused.pw_conv(xs[k], convs[k], fin=fin)[2]))
            torch.cuda.synchronize()
            for k, (sc, sh, _) in outs:
                assert torch.equal(sc, ref[k][0]) and torch.equal(sh, ref[k][1])


def test_arming_twice_is_refused_and_the_switch_falls_back():
    from p2p_bridge_amd import fused
    from p2p_bridge_amd._lib import lib

    gam = torch.ones(64, device="cuda")
    fin = (1024.0, 8, gam, gam, None, 1e-5, False)
    fused.arm_finisher(fin, 2, 64, gam.device)
    assert lib().p2pb_gn_finisher_armed() == 1
    with pytest.raises(RuntimeError):
        fused.arm_finisher(fin, 2, 64, gam.device)
    conv = torch.nn.Conv1d(16, 64, 1).cuda()
    with torch.no_grad():
        fused.pw_conv(torch.randn(2, 16, 1024, device="cuda"), conv)  # consumes the armed finisher
    assert lib().p2pb_gn_finisher_armed() == 0
    torch.cuda.synchronize()


def test_the_in_kernel_forms_give_the_same_bits():
    """P2PB_GN_FINISH=7 (read once per process): this file again with the split, wide and ping-pong kernels finishing their norms"""
    if os.environ.get("P2PB_GN_FINISH") == "7":
        pytest.skip("the re-run itself")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu"],
                       env=dict(os.environ, P2PB_GN_FINISH="7"), capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
