"""Room-level scoring (p2p_bridge_amd/evaluate_rooms.py <-> evaluate_rooms.py:21-308): the PLY reader (ASCII / binary,
extra vertex properties, polygon faces), the per-configuration metrics against float64 brute force and the oracle's
point-triangle distance, the scene-folder walk with its csv (already scored configurations are not scored again), the
reduction of over-complete predictions to the input scan's point count."""
import csv
import os
import struct

import numpy as np
import pytest
import torch

from oracle import cpu_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def er():
    from p2p_bridge_amd import evaluate_rooms
    return evaluate_rooms


def write_ply_ascii(path, pts, faces=None):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(pts))
        if faces is not None:
            f.write("element face %d\nproperty list uchar int vertex_indices\n" % len(faces))
        f.write("end_header\n")
        for p in pts:
            f.write("%.9g %.9g %.9g\n" % tuple(p))
        for t in faces or []:
            f.write("%d %s\n" % (len(t), " ".join(str(i) for i in t)))


def write_ply_binary(path, pts, colors, faces=None):
    with open(path, "wb") as f:
        h = "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n" % len(pts)
        h += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
        if faces is not None:
            h += "element face %d\nproperty list uchar uint vertex_indices\n" % len(faces)
        f.write((h + "end_header\n").encode())
        for p, c in zip(pts, colors):
            f.write(struct.pack("<dddBBB", *p, *c))
        for t in faces or []:
            f.write(struct.pack("<B" + "I" * len(t), len(t), *t))


CUBE_V = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]]
CUBE_Q = [[0, 1, 2, 3], [4, 5, 6, 7], [0, 1, 5, 4], [2, 3, 7, 6], [1, 2, 6, 5], [0, 3, 7, 4]]


def surface(n, seed, noise=0.0):
    g = np.random.RandomState(seed)
    p = g.rand(n, 3)
    p[np.arange(n), g.randint(0, 3, n)] = g.randint(0, 2, n)
    return p * 4 + [1, 2, 3] + noise * g.randn(n, 3)


def test_ply_reader(er, tmp_path):
    pts = surface(50, 0)
    cols = np.random.RandomState(1).randint(0, 255, (50, 3))
    write_ply_binary(str(tmp_path / "b.ply"), pts, cols, CUBE_Q)
    write_ply_ascii(str(tmp_path / "a.ply"), pts, [[0, 1, 2], [2, 3, 4, 5, 6]])
    b = er.read_ply(str(tmp_path / "b.ply"))
    assert b["points"].dtype == np.float64 and np.array_equal(b["points"], pts)
    assert b["faces"].shape == (12, 3) and b["faces"][:2].tolist() == [[0, 1, 2], [0, 2, 3]]
    a = er.read_ply(str(tmp_path / "a.ply"))
    assert np.allclose(a["points"], pts, rtol=1e-8) and a["faces"].tolist() == [[0, 1, 2], [2, 3, 4], [2, 4, 5], [2, 5, 6]]
    np.savetxt(str(tmp_path / "c.xyz"), np.concatenate([pts, cols], 1))
    assert np.allclose(er.read_cloud(str(tmp_path / "c.xyz")), pts) and np.array_equal(er.read_cloud(str(tmp_path / "b.ply")), pts)
    (tmp_path / "bad.ply").write_text("plx\n")
    with pytest.raises(ValueError):
        er.read_ply(str(tmp_path / "bad.ply"))


def brute(pred, gt):
    d = torch.cdist(torch.as_tensor(pred).double(), torch.as_tensor(gt).double()) ** 2
    return d.min(1).values.mean().item(), d.min(0).values.mean().item()


def test_scene_walk_and_metrics(er, tmp_path):
    scene = tmp_path / "root" / "scene0"
    (scene / "scans").mkdir(parents=True)
    model = scene / "predictions" / "ours"
    model.mkdir(parents=True)
    (scene / "predictions" / "iphone").mkdir()  # reserved names are not models
    iphone = surface(600, 1, 0.05)
    verts = (np.asarray(CUBE_V, float) * 4 + [1, 2, 3])
    write_ply_binary(str(scene / "scans" / "iphone.ply"), iphone, np.zeros((600, 3), int))
    write_ply_ascii(str(scene / "scans" / "mesh_aligned_0.05.ply"), verts, CUBE_Q)
    good, more, few = surface(600, 2, 0.01), surface(900, 3, 0.02), surface(100, 4)
    np.savetxt(str(model / "steps5.xyz"), good)
    write_ply_ascii(str(model / "steps10.ply"), more)  # more points than the scan: reduced by FPS to 600
    np.savetxt(str(model / "tiny.xyz"), few)           # fewer: skipped
    args = {"dataset": "snpp", "normalize": False, "suffix": ""}
    er.handle_scene(str(scene), args)
    path = str(model / "metrics.csv")
    rows = {r["model_config"]: r for r in csv.DictReader(open(path))}
    assert sorted(rows) == ["steps10", "steps5"] and list(next(iter(rows.values()))) == er.COLUMNS
    # the mesh "scan" has 8 vertices: Chamfer against them, point <-> triangle distances against the cube's 12 triangles
    a, b = brute(good, verts)
    assert abs(float(rows["steps5"]["cd_pred_gt"]) - a * 1000) < 1e-3 * a * 1000 and abs(float(rows["steps5"]["cd_gt_pred"]) - b * 1000) < 1e-3 * b * 1000
    tris = torch.as_tensor(verts).float()[torch.as_tensor(er.read_ply(str(scene / "scans" / "mesh_aligned_0.05.ply"))["faces"])]
    pd, _ = cpu_ops.point_face_dist(torch.as_tensor(good).float().contiguous(), tris.contiguous(), min_triangle_area=5e-3, which=0)
    fd, _ = cpu_ops.point_face_dist(torch.as_tensor(good).float().contiguous(), tris.contiguous(), min_triangle_area=5e-3, which=1)
    assert abs(float(rows["steps5"]["point_dist"]) - pd.double().mean().item() * 1000) < 1e-3 * float(rows["steps5"]["point_dist"]) + 1e-9
    assert abs(float(rows["steps5"]["face_dist"]) - fd.double().mean().item() * 1000) < 1e-3 * float(rows["steps5"]["face_dist"]) + 1e-9
    # the over-complete prediction was reduced to the scan's 600 points by exact FPS from its first point
    idx = cpu_ops.furthest_point_sampling_forward(torch.as_tensor(more).float().t().contiguous()[None], 600)[0].long()
    a, _ = brute(more[idx.numpy()], verts)
    assert abs(float(rows["steps10"]["cd_pred_gt"]) - a * 1000) < 1e-3 * a * 1000
    # a second pass scores nothing again (the file is unchanged); a new prediction is appended
    before = open(path).read()
    er.handle_scene(str(scene), args)
    assert open(path).read() == before
    np.savetxt(str(model / "steps20.xyz"), surface(600, 5, 0.03))
    er.handle_scene(str(scene), args)
    assert [r["model_config"] for r in csv.DictReader(open(path))] == ["steps5", "steps10", "steps20"] or \
        sorted(r["model_config"] for r in csv.DictReader(open(path))) == ["steps10", "steps20", "steps5"]
    # ARKit: no mesh -> the two Chamfer columns only; normalised metrics go to their own file
    sc2 = tmp_path / "root2" / "scene1"
    (sc2 / "scans").mkdir(parents=True)
    (sc2 / "predictions" / "ours").mkdir(parents=True)
    write_ply_ascii(str(sc2 / "scans" / "iphone.ply"), iphone)
    write_ply_ascii(str(sc2 / "scans" / "faro.ply"), surface(800, 7))
    np.savetxt(str(sc2 / "predictions" / "ours" / "p.xyz"), good)
    er.main(["--data_root", str(tmp_path / "root2"), "--dataset", "arkit", "--normalize"])
    r = list(csv.DictReader(open(str(sc2 / "predictions" / "ours" / "metrics.csv_normalized.csv"))))[0]
    assert r["point_dist"] == "" and r["face_dist"] == "" and float(r["cd_pred_gt"]) > 0
    faro = surface(800, 7)
    c = (faro.max(0) + faro.min(0)) / 2
    s = np.linalg.norm(faro - c, axis=1).max()
    a, b = brute((good - c) / s, (faro - c) / s)
    assert abs(float(r["cd_pred_gt"]) - a * 1000) < 2e-3 * a * 1000 and abs(float(r["cd_gt_pred"]) - b * 1000) < 2e-3 * b * 1000
