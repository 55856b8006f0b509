"""File formats and the summary table of p2p_bridge_amd/evaluation.py (no GPU): `.xyz` writer byte for byte
(utils/utils.py:5-10), the readers (models/evaluation.py:253-279), update_summary (models/evaluation.py:441-452)."""
import numpy as np
import torch

from p2p_bridge_amd import evaluation as ev


def test_xyz_and_off_round_trip(tmp_path):
    a = np.array([[1.5, -2.25, 3.0], [0.000001, 7.0, -8.125]])
    ev.write_array_to_xyz(str(tmp_path / "p.xyz"), a)
    assert (tmp_path / "p.xyz").read_text() == "%8f %8f %8f\n%8f %8f %8f" % tuple(a.ravel())
    assert torch.allclose(ev.load_xyz(str(tmp_path))["p"].double(), torch.from_numpy(a), atol=1e-6)
    (tmp_path / "single.xyz").write_text("1 2 3")  # one point still comes back as [1, 3]
    assert ev.load_xyz(str(tmp_path))["single"].shape == (1, 3)
    (tmp_path / "m.off").write_text("OFF\n# square + triangle\n5 2 0\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n2 2 2\n4 0 1 2 3\n3 0 1 4\n")
    m = ev.load_off(str(tmp_path))["m"]
    assert m["verts"].shape == (5, 3) and m["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]
    (tmp_path / "single.xyz").unlink()
    it = list(ev.input_iter(str(tmp_path)))
    assert {d["name"] for d in it} == {"p"}
    d = next(x for x in it if x["name"] == "p")
    assert abs(d["pcl_noisy"].norm(dim=1).max().item() - 1.0) < 1e-6  # on the unit sphere, centre / scale kept
    assert torch.allclose(d["pcl_noisy"] * d["scale"] + d["center"], torch.from_numpy(a).float(), atol=1e-5)


def test_update_summary_keeps_rows_and_columns(tmp_path):
    path = str(tmp_path / "s" / "Summary_X.csv")
    ev.update_summary(path, "m1", {"cd_sph(mean)": 0.25, "p2f(mean)": 1e-3})
    ev.update_summary(path, "m2", {"p2f(mean)": 2.0})
    t = ev.update_summary(path, "m1", {"cd_sph(mean)": 0.5})
    assert t["m1"] == {"cd_sph(mean)": "0.500000000000", "p2f(mean)": "0.001000000000"} and t["m2"]["p2f(mean)"] == "2.000000000000"
    assert open(path).read().split("\n")[0] == ",cd_sph(mean),p2f(mean)"
