"""Kernels of one stream must not depend on what another stream runs beside them. Round 4 found one that did: the
devoxelisation's corner staging (one wave writing all 64 points' rows with a compiler-merged ds_write_b96 / ds_write2_b32 pair)
returned wrong values for one wave's points when a matrix kernel of another stream shared the CU -- which is what the two-chain
graph sampler of bench.py does all the time (tools/dbg/devox_conc.py, chains_dbg2.py). These tests run the ops of one network
evaluation next to a stream of GEMM launches, and the two-chain sampler next to itself, and require BITWISE the serial results."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_devoxelize_beside_matrix_kernels_is_bitwise_stable():
    from p2p_bridge_amd import fused, pointnet2_batch_cuda as ext

    torch.manual_seed(0)
    B, N = 16, 8192
    xyz = torch.rand(B, 3, N, device="cuda") * 2 - 1
    conv = torch.nn.Conv1d(128, 128, 1).cuda()
    xp = torch.randn(B, 128, N, device="cuda")
    sc, sh = torch.rand(B, 128, device="cuda") + 0.5, torch.randn(B, 128, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.no_grad():
        for C, r in ((64, 32), (128, 16), (256, 8)):
            grid = torch.randn(B, r, r, r, C, device="cuda")
            vc, _ = ext.voxel_coords(xyz, r, True, 0.0)
            a, b = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
            f = lambda: fused.devoxelize_affine(grid, vc, r, a, b, channels_last=True)
            ref = f().clone()
            torch.cuda.synchronize()
            wrong = 0
            for _ in range(20):
                outs = []
                for _ in range(3):
                    with torch.cuda.stream(sa):
                        fused.pw_conv(xp, conv, sc, sh, swish=True)
                    with torch.cuda.stream(sb):
                        outs.append(f())
                torch.cuda.synchronize()
                wrong += sum(not torch.equal(o, ref) for o in outs)
            assert wrong == 0, (C, r, wrong)


def test_two_chain_sampler_equals_its_serial_replay():
    """the SAME two captured chain graphs replayed one after the other on one stream and side by side on two streams: bitwise
    equal (stock PVDS, 32 x 8192 points, 3 steps -- bench.py's configuration)"""
    import copy

    from oracle import net_ref
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet
    from test_host_logic import PVDS

    cfg = copy.deepcopy(PVDS)
    cfg["data"]["npoints"] = 8192
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
    model = product.build_model(cfg, sd, device="cuda")
    x, _ = net_ref.synthetic_patches(32, 8192, seed=5)
    x = x.cuda()
    model.sample_chains = 2
    run = lambda: model.sample(x_start=x, steps=3, log_count=3, verbose=False, graph=True)["x_chain"].clone()
    model._chains_serial = True
    ref = run()
    assert torch.equal(run(), ref)
    model._chains_serial = False
    for rep in range(5):
        c = run()
        d = (c - ref).abs().amax(dim=(2, 3))
        assert torch.equal(c, ref), (rep, [(int(b), int(k), float(d[b, k])) for b, k in (d > 0).nonzero().tolist()][:8])
