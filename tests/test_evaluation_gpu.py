"""Output side of the object pipeline (p2p_bridge_amd/evaluation.py <-> evaluate_objects.py, models/evaluation.py:206-452,
metrics/metrics.py:19-136, utils/utils.py:5-10): the `.xyz` / OFF readers and the writer, get_metrics in both branches
(Chamfer / EMD against brute-force float64, the oracle's approximate matching and scipy's optimal assignment, including
the reference's mean-of-chunk-means), the per-shape Evaluator on a synthetic data set against direct evaluations of the
same definitions (oracle point-triangle distance), the summary file, and the whole denoise -> write -> score loop
around the real sampler."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ev():
    from p2p_bridge_amd import evaluation
    return evaluation


CUBE_OFF = """OFF8 6 0
# a unit cube, quads
0 0 0
1 0 0
1 1 0
0 1 0
0 0 1
1 0 1   # comment after data
1 1 1
0 1 1

4 0 1 2 3
4 4 5 6 7
4 0 1 5 4
4 2 3 7 6
4 1 2 6 5
4 0 3 7 4
"""


def cube_cloud(n, seed, noise=0.0):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g)
    ax = torch.randint(0, 3, (n,), generator=g)
    side = torch.randint(0, 2, (n,), generator=g).float()
    p[torch.arange(n), ax] = side
    return p + noise * torch.randn(n, 3, generator=g)


def test_xyz_writer_and_readers(ev, tmp_path):
    a = np.array([[0.1234567891, -2.5, 3.0], [1e-7, 12345.678912, -0.000001]], dtype=np.float64)
    path = tmp_path / "a.xyz"
    ev.write_array_to_xyz(str(path), a)
    text = path.read_text()
    fmt = "\n".join([" ".join(["%8f"] * 3)] * 2)
    assert text == fmt % tuple(a.ravel()) and not text.endswith("\n")  # (utils/utils.py:5-10, byte for byte)
    got = ev.load_xyz(str(tmp_path))["a"]
    assert got.dtype == torch.float32 and torch.allclose(got.double(), torch.from_numpy(a), atol=6e-7, rtol=1e-6)
    (tmp_path / "cube.off").write_text(CUBE_OFF)
    (tmp_path / "tri.off").write_text("OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    meshes = ev.load_off(str(tmp_path))
    assert set(meshes) == {"cube", "tri"}
    v, f = meshes["cube"]["verts"], meshes["cube"]["faces"]
    assert v.shape == (8, 3) and f.shape == (12, 3) and f.dtype == torch.int64 and v[6].tolist() == [1, 1, 1]
    area = 0.5 * torch.linalg.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]).norm(dim=1).sum()
    assert abs(area.item() - 6.0) < 1e-6  # the fan covers every quad exactly
    assert meshes["tri"]["faces"].tolist() == [[0, 1, 2]]


def brute_cd(p, g, squared):
    d = torch.cdist(p.double(), g.double())
    if squared:
        d = d * d
    return d.min(2).values.mean(1) + d.min(1).values.mean(1)


def test_get_metrics_both_branches(ev):
    torch.manual_seed(0)
    B, N = 5, 256  # five clouds: one chunk of four + one of one (the reference averages per chunk first)
    gt = torch.rand(B, 3, N)
    pred = gt[:, :, torch.randperm(N)] + 0.02 * torch.randn(B, 3, N)

    class Loss:
        def loss(self, a, b):
            return ((a - b) ** 2).mean(dim=(1, 2))

    cd, emd, loss = ev.get_metrics(gt.cuda(), pred.cuda(), model=Loss(), fast=True)
    p3, g3 = pred.transpose(1, 2).contiguous(), gt.transpose(1, 2).contiguous()
    assert abs(cd - brute_cd(p3, g3, True).mean().item() * 1000) < 1e-3 * cd
    match = cpu_ops.approxmatch_forward(p3, g3)
    cost = cpu_ops.matchcost_forward(p3, g3, match) / N
    ref_emd = float(np.mean([cost[:4].mean().item(), cost[4:].mean().item()])) * 1000
    assert abs(emd - ref_emd) < 5e-3 * ref_emd, (emd, ref_emd)
    assert abs(loss - ((pred - gt) ** 2).mean().item()) < 1e-6
    # the layout does not matter: points-first input gives the same numbers
    cd2, emd2, _ = ev.get_metrics(g3.cuda(), p3.cuda(), model=None, fast=True)
    assert abs(cd2 - cd) < 1e-9 + 1e-6 * cd and abs(emd2 - emd) < 1e-6 * emd

    from scipy.optimize import linear_sum_assignment

    cd_s, emd_s, _ = ev.get_metrics(gt.cuda(), pred.cuda(), model=None, fast=False)
    assert abs(cd_s - brute_cd(p3, g3, False).mean().item() * 1000) < 1e-3 * cd_s
    best = []
    for i in range(B):
        d2 = torch.cdist(p3[i].double(), g3[i].double()) ** 2
        r, c = linear_sum_assignment(d2.numpy())
        best.append(float(np.sqrt(d2.numpy()[r, c].mean())))
    ref = float(np.mean(best)) * 1000
    assert ref * 0.999 <= emd_s <= ref * 1.05, (emd_s, ref)  # the auction is eps-optimal: never below, barely above


def make_dataset(root, ev, names, n_clean=2000, n_out=1500):
    os.makedirs(os.path.join(root, "data", "PUNet", "pointclouds", "test", "2000_poisson"))
    os.makedirs(os.path.join(root, "data", "PUNet", "meshes", "test"))
    os.makedirs(os.path.join(root, "out"))
    for i, name in enumerate(names):
        scale, shift = 1.0 + i, torch.tensor([0.3 * i, -0.2, 0.1])
        ev.write_array_to_xyz(os.path.join(root, "data", "PUNet", "pointclouds", "test", "2000_poisson", name + ".xyz"),
                              (cube_cloud(n_clean, i) * scale + shift).numpy())
        off = CUBE_OFF.split("\n")
        verts = torch.tensor([[float(t) for t in ln.split("#")[0].split()] for ln in off[2:10]])
        lines = ["OFF", "8 6 0"] + [" ".join("%.6f" % c for c in (v * scale + shift).tolist()) for v in verts] + off[11:17]
        with open(os.path.join(root, "data", "PUNet", "meshes", "test", name + ".off"), "w") as f:
            f.write("\n".join(lines) + "\n")
        ev.write_array_to_xyz(os.path.join(root, "out", name + ".xyz"),
                              (cube_cloud(n_out, 100 + i, noise=0.01 * (i + 1)) * scale + shift).numpy())


def test_evaluator_on_a_synthetic_dataset(ev, tmp_path):
    from p2p_bridge_amd import metrics as M

    root = str(tmp_path)
    names = ["a", "b", "c"]
    make_dataset(root, ev, names)
    ev.write_array_to_xyz(os.path.join(root, "out", "orphan.xyz"), cube_cloud(50, 9).numpy())  # no ground truth: skipped
    e = ev.Evaluator(os.path.join(root, "out"), os.path.join(root, "data"), "PUNet", os.path.join(root, "summary"), "run-1",
                     device="cuda", res_gts="2000_poisson")
    res = e.run()
    assert sorted(res) == names
    for name in names:
        up = ev.load_xyz(os.path.join(root, "out"))[name]
        high = ev.load_xyz(os.path.join(root, "data", "PUNet", "pointclouds", "test", "2000_poisson"))[name]
        mesh = ev.load_off(os.path.join(root, "data", "PUNet", "meshes", "test"))[name]
        # the definitions, evaluated independently in float64 / by the oracle's point-triangle distance
        hi, lo = high.max(0).values, high.min(0).values
        c = (hi + lo) / 2
        s = (high - c).norm(dim=1).max()
        cd = brute_cd(((up - c) / s)[None], ((high - c) / s)[None], True).item()
        assert abs(res[name]["cd_sph"] - cd) < 1e-4 * cd + 1e-9
        v = mesh["verts"]
        vc = (v.max(0).values + v.min(0).values) / 2
        vs = (v - vc).norm(dim=1).max()
        tris = ((v - vc) / vs)[mesh["faces"]]
        pts = ((up - vc) / vs).contiguous()
        pd, _ = cpu_ops.point_face_dist(pts, tris.contiguous(), min_triangle_area=0.0, which=0)
        fd, _ = cpu_ops.point_face_dist(pts, tris.contiguous(), min_triangle_area=0.0, which=1)
        ref = pd.double().mean().item() + fd.double().mean().item()
        assert abs(res[name]["p2f"] - ref) < 1e-4 * ref + 1e-9, (name, res[name]["p2f"], ref)
    # noisier results score worse; the summary holds the means with 12 decimals and survives a second model
    assert res["a"]["p2f"] < res["b"]["p2f"] < res["c"]["p2f"]
    path = os.path.join(root, "summary", "Summary_PUNet.csv")
    rows = [r.split(",") for r in open(path).read().strip().split("\n")]
    assert rows[0] == ["", "cd_sph(mean)", "p2f(mean)"] and rows[1][0] == "run-1"
    assert rows[1][1] == "%.12f" % np.mean([res[n]["cd_sph"] for n in names])
    ev.update_summary(path, "run-2", {"p2f(mean)": 0.5, "extra": 1.0})
    rows = [r.split(",") for r in open(path).read().strip().split("\n")]
    assert rows[0] == ["", "cd_sph(mean)", "p2f(mean)", "extra"] and rows[1][0] == "run-1" and rows[2] == ["run-2", "", "0.500000000000", "1.000000000000"]
    assert M.chamfer_distance_unit_sphere is not None


def test_denoise_write_score_loop_with_the_real_sampler(ev, tmp_path):
    from p2p_bridge_amd import p2pb as product

    g = os.path.join(os.path.dirname(__file__), "golden")
    cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
    w = np.load(os.path.join(g, "tiny_weights.npz"))
    model = product.build_model(cfg, {k: torch.from_numpy(w[k]).float() for k in w.files}, device="cuda:0")
    root = str(tmp_path)
    names = ["s0", "s1"]
    make_dataset(root, ev, names, n_clean=3000)
    noisy = os.path.join(root, "noisy", "PUNet_3000_poisson_0.01")
    os.makedirs(noisy)
    for i, name in enumerate(names):
        ev.write_array_to_xyz(os.path.join(noisy, name + ".xyz"), (cube_cloud(3000, 50 + i, noise=0.01) * (1 + i)).numpy())
    os.rename(os.path.join(root, "data", "PUNet", "pointclouds", "test", "2000_poisson"),
              os.path.join(root, "data", "PUNet", "pointclouds", "test", "3000_poisson"))
    out = ev.denoise_and_evaluate(model, {"use_ema": False, "steps": 2, "k": 2, "save_intermediate": True},
                                  data_path=os.path.join(root, "noisy"), dataset_root=os.path.join(root, "data"),
                                  output_root=os.path.join(root, "results"), resolutions=(3000,), noises=(0.01,),
                                  patch_size=1024)
    run = out["3000_0.01"]
    assert sorted(run) == names and all(np.isfinite(v) for r in run.values() for v in r.values())
    d = os.path.join(root, "results", "PUNet", "P2P-Bridge_steps_2_3000_0.01")
    for name in names:
        res = ev.load_xyz(os.path.join(d, "pcl"))[name]
        assert res.shape == (3000, 3)  # merged back to the input's point count, in the input's frame
        # the file holds exactly what the pipeline stages produce for that input (seeded random weights: the values are
        # not a denoised cube, the plumbing is what is checked): normalise -> patch_based_denoise -> de-normalise -> %8f
        from p2p_bridge_amd.denoise import patch_based_denoise
        from p2p_bridge_amd.punet_data import NormalizeUnitSphere

        src = torch.from_numpy(np.loadtxt(os.path.join(noisy, name + ".xyz")).astype(np.float32))
        unit, center, scale = NormalizeUnitSphere.normalize(src)
        again, _ = patch_based_denoise(model=model, pcl_noisy=unit.cuda(), patch_size=1024, seed_k=2,
                                       cfg={"use_ema": False, "steps": 2})
        again = again.cpu() * scale + center
        assert torch.isfinite(res).all() and (res - again).abs().max() <= 1e-6 + 1e-6 * again.abs().max()
        assert len(os.listdir(os.path.join(d, "steps", name))) == 2
    assert os.path.exists(os.path.join(d, "Summary_PUNet.csv"))


def test_in_training_evaluate(ev, tmp_path):
    """models/evaluation.py:76-203 without its pictures: the metrics of two accumulated validation batches equal
    get_metrics on the concatenated sampler outputs; the npy dumps hold what was scored"""
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd import train as T

    g = os.path.join(os.path.dirname(__file__), "golden")
    cfg = json.load(open(os.path.join(g, "tiny_cfg.json")))
    w = np.load(os.path.join(g, "tiny_weights.npz"))
    model = product.build_model(cfg, {k: torch.from_numpy(w[k]).float() for k in w.files}, device="cuda:0")
    cfg = dict(cfg, sampling={"accum_iter": 2}, out_sampling=str(tmp_path))
    cfg["data"] = dict(cfg["data"], dataset="PUNet")
    it = T.synthetic_punet_batches(2, 1000, seed=4, device="cuda")
    batches = [next(it) for _ in range(3)]  # the third is not consumed
    torch.manual_seed(0)
    m = ev.evaluate(model, batches, cfg, step=7, sampling=True, save_npy=True, fast=True)
    assert set(m) == {"cd", "emd", "mse", "cd_noisy", "emd_noisy", "mse_noisy"}
    pred, gt, noisy = (torch.from_numpy(np.load(os.path.join(str(tmp_path), "007_%s.npy" % n))) for n in ("pred", "gt", "noisy"))
    assert pred.shape == (4, 3, 896) and gt.shape == pred.shape  # two batches of two, 1000 -> 896 = 7 x 128 points
    want = torch.cat([b["clean_points"] for b in batches[:2]]).transpose(1, 2)[..., :896].cpu()
    assert torch.equal(gt, want)
    cd, emd, mse = ev.get_metrics(gt.cuda(), pred.cuda(), model=model, fast=True)
    assert abs(cd - m["cd"]) < 1e-6 * cd and abs(emd - m["emd"]) < 1e-5 * emd and abs(mse - m["mse"]) < 1e-6 * abs(mse) + 1e-9
    cdn, _, _ = ev.get_metrics(gt.cuda(), noisy.cuda(), model=model, fast=True)
    assert abs(cdn - m["cd_noisy"]) < 1e-6 * cdn
