"""Hand-written dense forward / backward for training (p2p_bridge_amd/dense.py, csrc/wgrad.hip): the 3x3x3 voxel
convolution and the k=1 convolutions as autograd Functions vs torch's own fp64 autograd of the same op."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


WTOL = {"bf16x3": 1e-4, "bf16x6": 1e-5, "fp32": 1e-5}  # P2PB_TRAIN_MATH: arithmetic of the weight-gradient GEMMs
# ... and of the data-gradient GEMMs (dense.dgrad_math): two bf16 terms / three products under "bf16x3" (<= 3 * 2^-18 per
# product; the reference's backward runs TF32, 2^-11), three terms / six products (fp32-faithful) otherwise
XTOL = {"bf16x3": 1e-4, "bf16x6": 5e-6, "fp32": 5e-6}


@pytest.mark.parametrize("math", ["bf16x3", "bf16x6", "fp32"])
@pytest.mark.parametrize("b,ci,co,r", [(2, 35, 32, 32), (2, 64, 64, 32), (3, 128, 64, 16), (2, 128, 128, 16),
                                       (2, 192, 128, 8), (1, 256, 256, 8), (2, 11, 8, 8), (2, 8, 16, 4),
                                       (3, 3, 70, 8)])
def test_conv3d_k3_forward_backward(b, ci, co, r, math, monkeypatch):
    from p2p_bridge_amd import dense

    monkeypatch.setenv("P2PB_TRAIN_MATH", math)
    torch.manual_seed(b * 1000 + ci + co + r)
    conv = nn.Conv3d(ci, co, 3, padding=1).cuda()
    x = torch.randn(b, ci, r, r, r, device="cuda", requires_grad=True)
    # sparse-ish input like a voxelised patch: most of the grid zero
    with torch.no_grad():
        x.mul_((torch.rand(b, 1, r, r, r, device="cuda") < 0.3).float())
    gy = torch.randn(b, co, r, r, r, device="cuda")
    y = dense.conv3d_k3(x, conv)
    y.backward(gy)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = conv.weight.detach().double().cpu().requires_grad_(True)
    b64 = conv.bias.detach().double().cpu().requires_grad_(True)
    y64 = F.conv3d(x64, w64, b64, padding=1)
    y64.backward(gy.double().cpu())
    assert _rel(y.detach(), y64.detach()) < 5e-6
    assert _rel(x.grad, x64.grad) < XTOL[math]
    assert _rel(conv.weight.grad, w64.grad) < WTOL[math]
    assert _rel(conv.bias.grad, b64.grad) < 5e-6


@pytest.mark.parametrize("math", ["bf16x3", "bf16x6", "fp32"])
@pytest.mark.parametrize("b,ci,co,shape", [(2, 3, 128, (1000,)), (2, 512, 1024, (2048,)), (3, 67, 64, (128, 32)),
                                           (2, 832, 256, (128,)), (8, 128, 3, (2048,)), (2, 35, 32, (512, 32)),
                                           (1, 256, 384, (8, 1)), (2, 320, 256, (8, 32))])
def test_pointwise_forward_backward(b, ci, co, shape, math, monkeypatch):
    from p2p_bridge_amd import dense

    monkeypatch.setenv("P2PB_TRAIN_MATH", math)
    torch.manual_seed(ci + co)
    conv = (nn.Conv1d if len(shape) == 1 else nn.Conv2d)(ci, co, 1).cuda()
    x = torch.randn(b, ci, *shape, device="cuda", requires_grad=True)
    gy = torch.randn(b, co, *shape, device="cuda")
    y = dense.pointwise(x, conv)
    assert y.shape == gy.shape
    y.backward(gy)
    x64 = x.detach().double().cpu().reshape(b, ci, -1).requires_grad_(True)
    w64 = conv.weight.detach().double().cpu().reshape(co, ci).requires_grad_(True)
    b64 = conv.bias.detach().double().cpu().requires_grad_(True)
    y64 = torch.einsum("oc,bcp->bop", w64, x64) + b64[None, :, None]
    y64.backward(gy.double().cpu().reshape(b, co, -1))
    assert _rel(y.detach().reshape(b, co, -1), y64.detach()) < 5e-6
    assert _rel(x.grad.reshape(b, ci, -1), x64.grad) < XTOL[math]
    assert _rel(conv.weight.grad.reshape(co, ci), w64.grad) < WTOL[math]
    assert _rel(conv.bias.grad, b64.grad) < 5e-6


@pytest.mark.parametrize("conv_math", ["fp32", "bf16x6", "f16x3"])
def test_conv3d_backward_under_every_conv_math(conv_math):
    """the data-gradient pass reads the forward weight through the ADJOINT pack, which exists only in the split
    form -- under P2PB_CONV_MATH=fp32 the backward used to raise. Forward + both gradients vs fp64 under each arithmetic."""
    from p2p_bridge_amd import dense, fused

    prev = fused.set_conv_math(conv_math)
    try:
        torch.manual_seed(5)
        conv = nn.Conv3d(64, 32, 3, padding=1).cuda()
        x = torch.randn(2, 64, 16, 16, 16, device="cuda", requires_grad=True)
        gy = torch.randn(2, 32, 16, 16, 16, device="cuda")
        y = dense.conv3d_k3(x, conv)
        y.backward(gy)
    finally:
        fused.set_conv_math(prev)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = conv.weight.detach().double().cpu().requires_grad_(True)
    y64 = F.conv3d(x64, w64, conv.bias.detach().double().cpu(), padding=1)
    y64.backward(gy.double().cpu())
    assert _rel(y.detach(), y64.detach()) < 5e-6
    assert _rel(x.grad, x64.grad) < XTOL["bf16x3"]  # (the default P2PB_TRAIN_MATH)
    assert _rel(conv.weight.grad, w64.grad) < 1e-4


def test_no_bias_and_no_input_grad():
    from p2p_bridge_amd import dense

    conv = nn.Conv2d(96, 384, 1, bias=False).cuda()
    x = torch.randn(2, 96, 32, 1, device="cuda")  # input does not require grad: only dW is computed
    y = dense.pointwise(x, conv)
    y.sum().backward()
    ref = x.reshape(2, 96, 32).sum(dim=(0, 2))[None, :].expand(384, -1)
    assert (conv.weight.grad.reshape(384, 96) - ref).abs().max().item() < 1e-4
    # weights updated in place (optimizer step): the cached transposed / packed copies follow
    x2 = torch.randn(2, 96, 32, 1, device="cuda", requires_grad=True)
    dense.pointwise(x2, conv).sum().backward()
    g1 = x2.grad.clone()
    with torch.no_grad():
        conv.weight.mul_(2.0)
    x2.grad = None
    dense.pointwise(x2, conv).sum().backward()
    assert torch.allclose(x2.grad, 2 * g1, rtol=1e-5, atol=1e-6)


def test_training_path_uses_no_miopen_convolutions():
    """one training forward+backward of the tiny network: every Conv3d / Conv1d / Conv2d goes through dense.py
    (checked by making torch's own convolution entry points raise)"""
    import json
    import os

    import numpy as np

    from conftest import GOLDEN
    from p2p_bridge_amd import p2pb as product

    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    w = np.load(os.path.join(GOLDEN, "tiny_weights.npz"))
    sd = {k: torch.from_numpy(w[k]).float() for k in w.files}
    run = np.load(os.path.join(GOLDEN, "tiny_run.npz"))
    model = product.build_model(cfg, sd, device="cuda")
    model.train()

    def boom(*a, **k):
        raise AssertionError("torch convolution called on the training path")

    saved = {n: getattr(F, n) for n in ("conv1d", "conv2d", "conv3d")}
    fsaved = {cls: cls.forward for cls in (nn.Conv1d, nn.Conv2d, nn.Conv3d)}
    try:
        for n in saved:
            setattr(F, n, boom)
        for cls in fsaved:
            cls.forward = boom
        loss = model(torch.from_numpy(run["clean"]).cuda(), torch.from_numpy(run["x_start"]).cuda(),
                     steps=torch.from_numpy(run["loss_steps"]))
        loss.backward()
    finally:
        for n, f in saved.items():
            setattr(F, n, f)
        for cls, f in fsaved.items():
            cls.forward = f
    assert abs(loss.item() - float(run["loss"])) <= 1e-4 * abs(float(run["loss"]))


@pytest.mark.parametrize("kind,b,ci,co,shape,swish", [
    ("adagn3d", 2, 16, 64, (8, 8, 8), True), ("adagn3d", 3, 8, 32, (16, 16, 16), False),
    ("adagn1d", 2, 67, 64, (1000,), True), ("adagn2d", 2, 35, 128, (64, 32), True), ("gn1d", 4, 64, 128, (2048,), True),
    ("mygn2d", 2, 3, 128, (512, 1), True), ("adagn3d", 2, 8, 16, (4, 4, 4), True), ("gn1d", 2, 3, 24, (333,), False)])
def test_conv_norm_act_forward_backward(kind, b, ci, co, shape, swish):
    """conv -> GroupNorm | AdaGN | MyGroupNorm -> [Swish] through dense.conv_norm_act (HIP conv emitting the statistics,
    folded norm, csrc/normact.hip backward) vs the same chain in torch fp64 autograd: output, dx, and every parameter
    gradient (conv weight / bias, norm gamma / beta, AdaGN's style Linear, the conditioning vector)"""
    from p2p_bridge_amd import dense
    from p2p_bridge_amd.pvcnn_unet import AdaGN, MyGroupNorm

    torch.manual_seed(len(kind) * 100 + co)
    conv = {"3d": nn.Conv3d(ci, co, 3, padding=1), "1d": nn.Conv1d(ci, co, 1), "2d": nn.Conv2d(ci, co, 1)}[kind[-2:]].cuda()
    if kind.startswith("adagn"):
        norm = AdaGN(co, 48, len(shape), 8).cuda()
        cond = torch.randn(b, 48, device="cuda", requires_grad=True)
    elif kind.startswith("mygn"):
        norm, cond = MyGroupNorm(32, co).cuda(), None
    else:
        norm, cond = nn.GroupNorm(8, co).cuda(), None
    gn = norm.norm if isinstance(norm, AdaGN) else (norm.group_norm if isinstance(norm, MyGroupNorm) else norm)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.normal_()
    x = torch.randn(b, ci, *shape, device="cuda", requires_grad=True)
    y = dense.conv_norm_act(x, conv, norm, cond, swish)
    gy = torch.randn_like(y)
    y.backward(gy)
    got = {"x": x.grad, "w": conv.weight.grad, "cb": conv.bias.grad, "gamma": gn.weight.grad, "beta": gn.bias.grad}
    if cond is not None:
        got.update(cond=cond.grad, ew=norm.emd.weight.grad, eb=norm.emd.bias.grad)
    # fp64 reference on the CPU
    d = lambda t: t.detach().double().cpu().requires_grad_(True)
    x64, w64, cb64, ga64, be64 = d(x), d(conv.weight), d(conv.bias), d(gn.weight), d(gn.bias)
    f = {"3d": lambda: F.conv3d(x64, w64, cb64, padding=1), "1d": lambda: F.conv1d(x64, w64, cb64),
         "2d": lambda: F.conv2d(x64, w64, cb64)}[kind[-2:]]
    h = F.group_norm(f(), gn.num_groups, ga64, be64, gn.eps)
    ref = {"x": x64, "w": w64, "cb": cb64, "gamma": ga64, "beta": be64}
    if cond is not None:
        c64, ew64, eb64 = d(cond), d(norm.emd.weight), d(norm.emd.bias)
        style = F.linear(c64, ew64, eb64).reshape(b, 2 * co, *([1] * len(shape)))
        fac, bia = style.chunk(2, 1)
        h = h * fac + bia
        ref.update(cond=c64, ew=ew64, eb=eb64)
    y64 = h * torch.sigmoid(h) if swish else h
    y64.backward(gy.double().cpu())
    assert _rel(y.detach(), y64.detach()) < 1e-5
    for k, v in ref.items():
        tol = 2e-4 if k in ("w",) else 5e-5  # (conv weight gradient: bf16x3 operands by default)
        if k == "cb":  # sums of dx over a channel: heavy cancellation (they add up to zero over every group)
            err = (got[k].double().cpu() - v.grad).abs().max().item()
            assert err < 1e-5 * max(1.0, got["x"].abs().max().item() * x[0, 0].numel() ** 0.5), (k, err)
            continue
        assert _rel(got[k], v.grad) < tol, (k, _rel(got[k], v.grad))


@pytest.mark.parametrize("b,ci,co,r,n", [(3, 35, 32, 32, 2048), (2, 64, 64, 32, 1500), (4, 128, 64, 16, 512), (2, 24, 40, 16, 300),
                                         (1, 8, 8, 32, 1)])
def test_sparse_weight_gradient_of_a_first_convolution(b, ci, co, r, n):
    """round 5 (csrc/wgrad.hip conv3d_k3_wgrad_occ_kernel): the weight gradient of a PVConv's first convolution over the occupied
    voxels only -- x = avg_voxelize(features) is zero elsewhere -- against torch's fp64 convolution backward and against the dense
    kernel; grid-boundary voxels, ragged channel counts, a single point"""
    from p2p_bridge_amd import dense, layers as L

    torch.manual_seed(b * 1000 + ci)
    feats = torch.randn(b, ci, n, device="cuda", requires_grad=True)
    vox = torch.randint(0, r, (b, 3, n), device="cuda", dtype=torch.int32)
    vox[:, :, : max(1, n // 8)] = torch.randint(0, 2, (b, 3, max(1, n // 8)), device="cuda", dtype=torch.int32) * (r - 1)  # corners / faces
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    gy = torch.randn(b, co, r, r, r, device="cuda")
    grads = {}
    for sparse in (True, False):
        dense.SPARSE_WGRAD = sparse
        try:
            x = L.avg_voxelize(feats, vox, r)
            assert getattr(x, "_p2pb_occ", None) is not None
            y = dense.conv3d_k3(x, conv)
            gw, gb, gf = torch.autograd.grad(y, [conv.weight, conv.bias, feats], gy)
        finally:
            dense.SPARSE_WGRAD = True
        grads[sparse] = (gw, gb, gf)
    x64 = L.avg_voxelize(feats, vox, r).detach().double()
    w64 = conv.weight.detach().double().requires_grad_(True)
    b64 = conv.bias.detach().double().requires_grad_(True)
    y64 = torch.nn.functional.conv3d(x64, w64, b64, padding=1)
    rw, rb = torch.autograd.grad(y64, [w64, b64], gy.double())
    gw, gb, gf = grads[True]
    scale = rw.abs().max().item() + 1e-30
    assert (gw.double() - rw).abs().max().item() < 2e-6 * scale * max(1.0, (n * b) ** 0.5 / 10), (gw.double() - rw).abs().max().item() / scale
    assert torch.allclose(gb.double(), rb, rtol=1e-4, atol=1e-3 * rb.abs().max().item())
    # the dense kernel (bf16x3 by default) agrees to its own precision; the data gradient is the same launch either way
    assert (grads[False][0].double() - rw).abs().max().item() < 2e-2 * scale
    assert torch.equal(gf, grads[False][2])


@pytest.mark.parametrize("edit", ["inplace_add", "inplace_dropout", "explicit_stale_occ", "untouched"])
def test_sparse_weight_gradient_is_dropped_when_the_grid_was_edited(edit):
    """the occupied-voxel weight gradient assumes x == 0 wherever counts == 0. The occupancy record is tied to the
    grid's storage pointer and version counter, so a drop-in caller that edits the grid between the voxelisation and the
    convolution (in place: the Python attribute survives) gets the DENSE weight gradient -- the right answer -- instead of a
    silently wrong one; an untouched grid still takes the sparse kernel"""
    from p2p_bridge_amd import dense, layers as L

    torch.manual_seed(5)
    b, ci, co, r, n = 2, 16, 16, 16, 300
    feats = torch.randn(b, ci, n, device="cuda")
    vox = torch.randint(0, r, (b, 3, n), device="cuda", dtype=torch.int32)
    conv = torch.nn.Conv3d(ci, co, 3, padding=1).cuda()
    gy = torch.randn(b, co, r, r, r, device="cuda")
    x = L.avg_voxelize(feats, vox, r)
    occ = L.occupancy_of(x)
    assert occ is not None and occ.describes(x)
    if edit == "inplace_add":
        x.add_(0.5)  # non-zero everywhere now; `_p2pb_occ` is still attached to the tensor object
    elif edit == "inplace_dropout":
        torch.nn.functional.dropout(x.add_(1.0), 0.5, training=True, inplace=True)
    elif edit == "explicit_stale_occ":
        x = x + 0.25  # another tensor; the caller hands the OLD record over explicitly
    if edit != "untouched":
        assert not occ.describes(x) and (edit == "explicit_stale_occ" or L.occupancy_of(x) is None)
    used = {}
    real = dense.call

    def spy(name, *a):
        used[name] = used.get(name, 0) + 1
        return real(name, *a)

    dense.call = spy
    try:
        y = dense.conv3d_k3(x, conv, occ=occ if edit == "explicit_stale_occ" else None)
        (gw,) = torch.autograd.grad(y, [conv.weight], gy)
    finally:
        dense.call = real
    assert ("p2pb_conv3d_k3_wgrad_occ" in used) == (edit == "untouched"), used
    w64 = conv.weight.detach().double().requires_grad_(True)
    y64 = torch.nn.functional.conv3d(x.detach().double(), w64, conv.bias.detach().double(), padding=1)
    (rw,) = torch.autograd.grad(y64, [w64], gy.double())
    tol = (2e-5 if edit == "untouched" else 2e-2) * rw.abs().max().item()  # (dense form: bf16x3 products by default)
    assert (gw.double() - rw).abs().max().item() < tol


@pytest.mark.parametrize("b,c", [(8, 64), (3, 256), (2, 32), (5, 1024)])
def test_se_gate_forward_backward(b, c):
    """dense.se_gate (csrc/normact.hip se_gate_*): SE3d's excitation and every gradient vs the module's own fp64 autograd"""
    from p2p_bridge_amd import dense
    from p2p_bridge_amd.pvcnn_unet import SE3d

    torch.manual_seed(c + b)
    se = SE3d(c).cuda()
    mean = torch.randn(b, c, device="cuda", requires_grad=True)
    dg = torch.randn(b, c, device="cuda")
    g = dense.se_gate(mean, se.fc)
    g.backward(dg)
    m64 = mean.detach().double().cpu().requires_grad_(True)
    w1, w2 = (se.fc[i].weight.detach().double().cpu().requires_grad_(True) for i in (0, 2))
    g64 = torch.sigmoid(torch.relu(m64 @ w1.t()) @ w2.t())
    g64.backward(dg.double().cpu())
    assert _rel(g.detach(), g64.detach()) < 1e-6
    assert _rel(mean.grad, m64.grad) < 1e-5
    assert _rel(se.fc[0].weight.grad, w1.grad) < 1e-5
    assert _rel(se.fc[2].weight.grad, w2.grad) < 1e-5


@pytest.mark.parametrize("shape", [(8, 64, 512, 32), (3, 128, 128, 32), (2, 1024, 2048), (5, 7, 33), (4, 9, 100), (2, 3, 1), (1, 2, 300, 6)])
def test_row_max_forward_backward(shape):
    """dense.row_max (csrc/normact.hip, round 6): the set abstractions' neighbour max and Pnet2Stage's global max-pools in train()
    -- values and gradients equal torch's x.max(-1) (first index on exact ties, like torch's index-based backward), ragged and
    unaligned rows, a NaN in a row reaches the output"""
    from p2p_bridge_amd import dense

    torch.manual_seed(sum(shape))
    x = torch.randn(*shape, device="cuda")
    x[..., 0, :] = x[..., 0, :].round()  # exact ties in the first row of every slab
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = dense.row_max(xa)
    vb, ib = xb.max(dim=-1)
    assert torch.equal(ya, vb)
    gy = torch.randn_like(ya)
    ya.backward(gy)
    # reference gradient with the FIRST maximal index (torch's CUDA reduction may pick any of several equal maxima)
    first = (x == vb.unsqueeze(-1)).float().argmax(dim=-1)
    want = torch.zeros_like(x).scatter_(-1, first.unsqueeze(-1), gy.unsqueeze(-1))
    assert torch.equal(xa.grad, want)
    assert xa.grad.sum().item() == pytest.approx(gy.sum().item(), rel=1e-5, abs=1e-4)
    xn = x.clone()
    xn.view(-1, shape[-1])[0, shape[-1] // 2] = float("nan")
    assert torch.isnan(dense.row_max(xn).view(-1)[0]) and torch.isfinite(dense.row_max(xn).view(-1)[1:]).all()


def _fold_case(kind, b, ci, co, shape):
    from p2p_bridge_amd.pvcnn_unet import AdaGN

    conv = {"3d": nn.Conv3d(ci, co, 3, padding=1), "1d": nn.Conv1d(ci, co, 1)}[kind[-2:]].cuda()
    if kind.startswith("adagn"):
        norm = AdaGN(co, 48, len(shape), 8).cuda()
        cond = torch.randn(b, 48, device="cuda", requires_grad=True)
    else:
        norm, cond = nn.GroupNorm(8, co).cuda(), None
    gn = norm.norm if isinstance(norm, AdaGN) else norm
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.normal_()
    x = torch.randn(b, ci, *shape, device="cuda", requires_grad=True)
    return conv, norm, gn, cond, x


def _fp64_chain(kind, conv, norm, gn, cond, x, swish, shape):
    """fp64 leaves + the conv -> norm -> [Swish] value on the CPU"""
    from p2p_bridge_amd.pvcnn_unet import AdaGN

    d = lambda t: t.detach().double().cpu().requires_grad_(True)  # noqa: E731
    leaves = {"x": d(x), "w": d(conv.weight), "cb": d(conv.bias), "gamma": d(gn.weight), "beta": d(gn.bias)}
    h = F.conv3d(leaves["x"], leaves["w"], leaves["cb"], padding=1) if kind.endswith("3d") else F.conv1d(leaves["x"], leaves["w"], leaves["cb"])
    h = F.group_norm(h, gn.num_groups, leaves["gamma"], leaves["beta"], gn.eps)
    if isinstance(norm, AdaGN):
        leaves.update(cond=d(cond), ew=d(norm.emd.weight), eb=d(norm.emd.bias))
        fac, bia = F.linear(leaves["cond"], leaves["ew"], leaves["eb"]).reshape(x.shape[0], -1, *([1] * len(shape))).chunk(2, 1)
        h = h * fac + bia
    return leaves, (h * torch.sigmoid(h) if swish else h)


def _got(conv, norm, gn, cond, x):
    from p2p_bridge_amd.pvcnn_unet import AdaGN

    got = {"x": x.grad, "w": conv.weight.grad, "gamma": gn.weight.grad, "beta": gn.bias.grad}
    if isinstance(norm, AdaGN):
        got.update(cond=cond.grad, ew=norm.emd.weight.grad, eb=norm.emd.bias.grad)
    return got


@pytest.mark.parametrize("kind,b,ci,co,shape,gated", [("adagn1d", 2, 32, 64, (2048,), True), ("gn1d", 3, 35, 32, (333,), True),
                                                      ("adagn1d", 2, 16, 128, (512,), False), ("adagn3d", 2, 8, 16, (8, 8, 8), True)])
def test_folded_residual_and_gate(kind, b, ci, co, shape, gated):
    """y = Swish(norm(conv x)) + residual [* rgate] inside the folded norm's launches (PVConv: point branch + devoxelised grid *
    SE gate) vs fp64 autograd of the same expression: value and every gradient, the residual's and the gate's included"""
    from p2p_bridge_amd import dense

    torch.manual_seed(co + len(shape))
    conv, norm, gn, cond, x = _fold_case(kind, b, ci, co, shape)
    res = torch.randn(b, co, *shape, device="cuda", requires_grad=True)
    gate = torch.rand(b, co, device="cuda", requires_grad=True) if gated else None
    y = dense.conv_norm_act(x, conv, norm, cond, True, residual=res, rgate=gate)
    gy = torch.randn_like(y)
    y.backward(gy)
    got = _got(conv, norm, gn, cond, x)
    got["res"] = res.grad
    leaves, h = _fp64_chain(kind, conv, norm, gn, cond, x, True, shape)
    leaves["res"] = res.detach().double().cpu().requires_grad_(True)
    r = leaves["res"]
    if gated:
        got["gate"] = gate.grad
        leaves["gate"] = gate.detach().double().cpu().requires_grad_(True)
        r = r * leaves["gate"].reshape(b, co, *([1] * len(shape)))
    y64 = h + r
    y64.backward(gy.double().cpu())
    assert _rel(y.detach(), y64.detach()) < 1e-5
    for k, v in leaves.items():
        if k == "cb":
            continue
        assert _rel(got[k], v.grad) < (2e-4 if k == "w" else 5e-5), (k, _rel(got[k], v.grad))


@pytest.mark.parametrize("kind,b,ci,co,shape", [("adagn3d", 2, 16, 64, (8, 8, 8)), ("adagn3d", 3, 8, 32, (16, 16, 16)), ("gn1d", 2, 24, 40, (300,))])
def test_folded_channel_mean(kind, b, ci, co, shape):
    """(y, mean of y over the positions) from ONE pass -- the mean out of the convolution's statistics, its gradient folded into the
    norm's backward (SE3d's squeeze without a grid-sized mean / div / add) -- vs fp64 autograd of y and y.mean over the grid"""
    from p2p_bridge_amd import dense

    torch.manual_seed(co)
    conv, norm, gn, cond, x = _fold_case(kind, b, ci, co, shape)
    y, m = dense.conv_norm_act(x, conv, norm, cond, False, want_mean=True)
    assert m.shape == (b, co)
    gy, gm = torch.randn_like(y), torch.randn_like(m) * 30
    torch.autograd.backward([y, m], [gy, gm])
    got = _got(conv, norm, gn, cond, x)
    leaves, h = _fp64_chain(kind, conv, norm, gn, cond, x, False, shape)
    m64 = h.reshape(b, co, -1).mean(-1)
    torch.autograd.backward([h, m64], [gy.double().cpu(), gm.double().cpu()])
    assert _rel(y.detach(), h.detach()) < 1e-5 and (m.detach().double().cpu() - m64.detach()).abs().max().item() < 2e-6
    for k, v in leaves.items():
        if k == "cb":
            continue
        assert _rel(got[k], v.grad) < (2e-4 if k == "w" else 5e-5), (k, _rel(got[k], v.grad))
    # only the mean is used (no gradient reaches y)
    for t in (x, conv.weight, gn.weight):
        t.grad = None
    y2, m2 = dense.conv_norm_act(x, conv, norm, cond, False, want_mean=True)
    m2.backward(gm)
    leaves2, h2 = _fp64_chain(kind, conv, norm, gn, cond, x, False, shape)
    h2.reshape(b, co, -1).mean(-1).backward(gm.double().cpu())
    assert _rel(x.grad, leaves2["x"].grad) < 5e-5


@pytest.mark.parametrize("kind,b,ci,co,shape,p", [("adagn3d", 2, 8, 32, (16, 16, 16), 0.1), ("gn1d", 2, 64, 128, (2048,), 0.15),
                                                  ("adagn3d", 2, 8, 16, (8, 8, 8), 0.5), ("gn1d", 2, 24, 40, (333,), 0.1)])
def test_folded_dropout(kind, b, ci, co, shape, p):
    """nn.Dropout(p) behind the Swish inside the folded norm's launches: every element is 0 or the undropped value / (1 - p), the
    kept fraction is 1 - p to 4 sigma, the backward pass regenerates the same mask (every gradient equals fp64 autograd of the
    chain with the mask read off the forward pass), the same (seed, salt) replays the mask, another seed / salt draws another"""
    from p2p_bridge_amd import dense

    torch.manual_seed(co)
    conv, norm, gn, cond, x = _fold_case(kind, b, ci, co, shape)
    seed = dense.dropout_seed(x.device)
    plain = dense.conv_norm_act(x, conv, norm, cond, True).detach()
    y = dense.conv_norm_act(x, conv, norm, cond, True, dropout=(p, seed, 3))
    keep = y.detach() != 0
    scale = torch.tensor(1.0, device="cuda") / (1.0 - torch.tensor(p, dtype=torch.float32, device="cuda"))
    assert torch.equal(y.detach()[keep], (plain * scale)[keep])
    dropped = ~keep & (plain != 0)
    n = y.numel()
    assert abs(dropped.sum().item() / n - p) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-6
    # per-row fractions too (a hash that only mixed the low index bits would pass the global count)
    rows = dropped.reshape(b * co, -1).float().mean(1)
    assert (rows - p).abs().max().item() < 6 * (p * (1 - p) / rows.numel() ** 0 / dropped[0, 0].numel()) ** 0.5 + 1e-6
    gy = torch.randn_like(y)
    y.backward(gy)
    got = _got(conv, norm, gn, cond, x)
    leaves, h = _fp64_chain(kind, conv, norm, gn, cond, x, True, shape)
    (h * (~dropped).double().cpu() * float(scale)).backward(gy.double().cpu())
    for k, v in leaves.items():
        if k == "cb":
            continue
        assert _rel(got[k], v.grad) < (2e-4 if k == "w" else 5e-5), (k, _rel(got[k], v.grad))
    again = dense.conv_norm_act(x, conv, norm, cond, True, dropout=(p, seed, 3)).detach()
    assert torch.equal(again, y.detach())
    other_salt = dense.conv_norm_act(x, conv, norm, cond, True, dropout=(p, seed, 4)).detach()
    other_seed = dense.conv_norm_act(x, conv, norm, cond, True, dropout=(p, dense.dropout_seed(x.device), 3)).detach()
    for o in (other_salt, other_seed):
        both = ((o == 0) & dropped).sum().item() / n  # independent masks: P(both dropped) = p^2
        assert abs(both - p * p) < 5 * (p * p * (1 - p * p) / n) ** 0.5 + 1e-6


def test_dropout_seed_follows_the_generator_and_replays_fresh_in_a_graph():
    from p2p_bridge_amd import dense

    torch.manual_seed(5)
    a = dense.dropout_seed("cuda")
    torch.manual_seed(5)
    assert torch.equal(a, dense.dropout_seed("cuda")) and not torch.equal(a, dense.dropout_seed("cuda"))
    torch.manual_seed(1)
    conv, norm, gn, cond, x = _fold_case("gn1d", 2, 16, 32, (256,))
    x = x.detach()
    out = torch.empty(2, 32, 256, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(2):
            out.copy_(dense._NormAct.apply(*_norm_args(dense, conv, gn, x), True, None, None, (0.3, dense.dropout_seed("cuda"), 1), False)[0])
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out.copy_(dense._NormAct.apply(*_norm_args(dense, conv, gn, x), True, None, None, (0.3, dense.dropout_seed("cuda"), 1), False)[0])
    g.replay()
    m1 = out == 0
    g.replay()
    m2 = out == 0
    assert 0.2 < m1.float().mean().item() < 0.4 and not torch.equal(m1, m2)


def _norm_args(dense, conv, gn, x):
    y, st = dense.pointwise(x, conv, True)
    return y, gn.weight, gn.bias, None, st, gn.num_groups, gn.eps


def test_backward_reads_a_channel_slice_of_a_concatenation_in_place():
    """the gradient that reaches a layer through torch.cat is a channel slice of a wider tensor (sample-pitched, not contiguous):
    the folded norm's backward, the grouping adjoint and the 3-NN interpolation adjoint read it in place -- same results as from a
    contiguous copy (the LDS scatter kernels add in a different order: 1e-6), and no copy is made"""
    from p2p_bridge_amd import dense
    from p2p_bridge_amd import pointnet2_batch_cuda as ext

    torch.manual_seed(2)
    big = torch.randn(4, 96 + 35, 128, 32, device="cuda")
    g = big[:, 35:]
    assert not g.is_contiguous() and ext.sample_pitch(g) == big.stride(0) and ext.sample_pitch(big) == big.stride(0)
    assert ext.sample_pitch(big[:, :, ::2]) is None and ext.sample_pitch(big.transpose(1, 2)) is None
    idx = torch.randint(0, 512, (4, 128, 32), device="cuda", dtype=torch.int32)
    a, b = ext.grouping_backward_pitched(g, idx, 512), ext.grouping_backward(g.contiguous(), idx, 512)
    assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item()
    big3 = torch.randn(3, 64 + 200, 2048, device="cuda")
    g3 = big3[:, :200]
    idx3 = torch.randint(0, 512, (3, 3, 2048), device="cuda", dtype=torch.int32)
    w3 = torch.rand(3, 3, 2048, device="cuda")
    a, b = ext.three_nearest_neighbors_interpolate_backward_pitched(g3, idx3, w3, 512), \
        ext.three_nearest_neighbors_interpolate_backward(g3.contiguous(), idx3, w3, 512)
    assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item()
    # the folded norm: y -> cat -> loss, against the same with the slice copied first (bit-identical: fixed summation orders)
    for shape, p in (((512,), 0.0), ((333,), 0.0), ((8, 8, 8), 0.2)):
        conv, norm, gn, cond, x = _fold_case("adagn3d" if len(shape) == 3 else "adagn1d", 2, 16, 32, shape)
        seed = dense.dropout_seed("cuda")
        other = torch.randn(2, 5, *shape, device="cuda")
        gz = torch.randn(2, 37, *shape, device="cuda")
        res = []
        for copy in (False, True):
            for t in (x, conv.weight, gn.weight, cond):
                t.grad = None
            y = dense.conv_norm_act(x, conv, norm, cond, True, dropout=(p, seed, 1) if p else None)
            if copy:
                y.backward(gz[:, 5:].contiguous())
            else:
                torch.cat([other, y], 1).backward(gz)
            res.append([t.grad.clone() for t in (x, conv.weight, gn.weight, cond)])
        for u, v in zip(*res):
            assert torch.equal(u, v)


@pytest.mark.parametrize("b,c,n,m,u", [(8, 32, 2048, 512, 32), (3, 64, 512, 128, 32), (2, 320, 32, 8, 32), (2, 5, 300, 77, 16)])
def test_group_concat_one_launch_equals_the_reference_graph(b, c, n, m, u):
    """BallQuery's [grouped coordinates - centres | grouped features] as ONE autograd node (layers.GroupConcat) against the
    reference's graph of two groupings, a subtraction and a concatenation: the same bits forward, the same feature gradient
    (LDS scatter order: 1e-6)"""
    from p2p_bridge_amd import layers as L
    from p2p_bridge_amd.pvcnn_unet import BallQuery

    torch.manual_seed(n)
    pts = torch.rand(b, 3, n, device="cuda")
    cen = pts[:, :, :m].contiguous()
    feat = torch.randn(b, c, n, device="cuda", requires_grad=True)
    bq = BallQuery(0.3, u)
    y = bq(pts, cen, feat)
    assert y.grad_fn.__class__.__name__.startswith("GroupConcat")
    idx = L.ball_query(cen, pts, 0.3, u)
    f2 = feat.detach().clone().requires_grad_(True)
    ref = torch.cat([L.pvcnn_grouping(pts, idx) - cen.unsqueeze(-1), L.pvcnn_grouping(f2, idx)], dim=1)
    assert torch.equal(y, ref)
    g = torch.randn_like(y)
    y.backward(g)
    ref.backward(g)
    assert (feat.grad - f2.grad).abs().max().item() <= 1e-6 * f2.grad.abs().max().item()
    # coordinates that require a gradient keep the unfused operators
    pts_g = pts.clone().requires_grad_(True)
    assert not bq(pts_g, cen, feat).grad_fn.__class__.__name__.startswith("GroupConcat")
