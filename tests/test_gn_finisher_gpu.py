"""The GroupNorm that follows a 1x1 convolution, handed to the convolution's own entry point (p2pb_gn_finisher_arm, csrc/common.h
GnFinish): scale / shift / channel mean must be the SAME BITS as the separate p2pb_gn_affine_params launch gives on the same
partials (the entry point puts the gn_affine launch behind the producer; round 4's in-kernel forms left the library in round 5) --
with and without AdaGN styles, ragged channel tiles, repeated launches, two streams side by side; a shape the norm cannot take is
refused."""
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


@pytest.mark.parametrize("b,ci,co,p,groups", SHAPES)
def test_finished_by_the_producer_equals_the_separate_launch(b, ci, co, p, groups):
    for rep in range(3):
        _run(b, ci, co, p, groups, styled=rep != 1, want_mean=rep == 2, seed=rep)


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


def test_arming_twice_is_refused():
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


def test_a_shape_the_norm_cannot_take_is_refused():
    """cout % groups != 0 (MyGroupNorm(32, cout) with a ragged cout), cout / groups > 256 or a style row shorter than
    2 * cout used to run the finisher with a truncated group / past its LDS table; the entry point returns P2PB_EINVAL instead"""
    from p2p_bridge_amd import fused
    from p2p_bridge_amd._lib import P2PBError, lib

    x = torch.randn(2, 16, 1024, device="cuda")
    for co, groups, style_cols in ((40, 32, None), (1040, 4, None), (64, 8, 100)):
        conv = torch.nn.Conv1d(16, co, 1).cuda()
        gam = torch.ones(co, device="cuda")
        style = None if style_cols is None else torch.randn(2, style_cols, device="cuda")
        with torch.no_grad(), pytest.raises((P2PBError, RuntimeError)):
            fused.pw_conv(x, conv, fin=(1024.0, groups, gam, gam, style, 1e-5, False))
        lib().p2pb_gn_finisher_disarm()
        torch.cuda.synchronize()
