"""Room-level scoring: what `evaluate_rooms.py` does with the results of `denoise_room.py` -- for every scene folder
`<root>/<scene>/{scans/, predictions<suffix>/<model>/*.{ply,xyz}}` the two one-sided Chamfer distances of each predicted
cloud against the Faro scan and, for ScanNet++ (which ships a mesh), the point-to-face / face-to-point distances, all
x 1000, appended to `<scene>/metrics/<model>/metrics<suffix>.csv` (already scored configurations are skipped).
On this package's HIP metrics (`metrics.cd_unit_sphere`, `metrics.point_face_dist`: csrc/chamfer.hip, p2m.hip).

  get_mectrics (sic)        evaluate_rooms.py:21-66      (the reference's spelling is kept; `get_metrics` is an alias)
  calculate_model_metrics   :69-98
  load_folder_snpp / load_folder_arkit   :101-231       (scans/iphone<suffix>.ply, scans/mesh_aligned_0.05.ply | scans/faro.ply)
  handle_scene, main        :234-308

Third-party edges, restated ("parity unpinned": none of them is under /root/reference):
  * open3d's readers -> `read_ply` below: ASCII and binary PLY, vertex positions + faces (polygons fanned);
  * fpsample.bucket_fps_kdline_sampling (predictions with MORE points than the iPhone scan are reduced to its count) ->
    this package's exact FPS from point 0;
  * pandas -> csv text with the reference's column order.
The reference passes a `segments` argument that its own `get_mectrics` does not take (:93) and reads a `segments` entry
that its loaders never set (:80); neither exists here.
"""
import argparse
import csv
import os
import struct
from typing import Dict, Optional

import numpy as np
import torch

from . import metrics as M

MULTIPLIER = 10 ** 3
COLUMNS = ["model_config", "point_dist", "face_dist", "cd_pred_gt", "cd_gt_pred"]
_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_ply(path: str) -> Dict[str, Optional[np.ndarray]]:
    """-> {"points": f64[N,3], "faces": i64[F,3] | None}; ascii, binary_little_endian and binary_big_endian"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []  # elements: [name, count, [(kind, ...)]]
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                elements[-1][2].append(("list", tok[2], tok[3], tok[4]) if tok[1] == "list" else ("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        out = {"points": None, "faces": None}
        if fmt == "ascii":
            words = f.read().split()
            pos = 0
            for name, count, props in elements:
                rows = []
                for _ in range(count):
                    row = []
                    for p in props:
                        if p[0] == "scalar":
                            row.append(float(words[pos]))
                            pos += 1
                        else:
                            k = int(words[pos])
                            row.append([int(w) for w in words[pos + 1:pos + 1 + k]])
                            pos += 1 + k
                    rows.append(row)
                _collect(out, name, props, rows)
            return out
        end = "<" if fmt == "binary_little_endian" else ">"
        for name, count, props in elements:
            if all(p[0] == "scalar" for p in props):  # fixed-size records: one structured read
                dt = np.dtype([(p[2], end + _PLY_TYPES[p[1]]) for p in props])
                rec = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                if name == "vertex":
                    out["points"] = np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float64)
                continue
            rows = []
            for _ in range(count):
                row = []
                for p in props:
                    if p[0] == "scalar":
                        c = _PLY_TYPES[p[1]]
                        row.append(struct.unpack(end + c, f.read(struct.calcsize(c)))[0])
                    else:
                        ck, ci = _PLY_TYPES[p[1]], _PLY_TYPES[p[2]]
                        k = struct.unpack(end + ck, f.read(struct.calcsize(ck)))[0]
                        row.append(list(struct.unpack(end + ci * k, f.read(struct.calcsize(ci) * k))))
                rows.append(row)
            _collect(out, name, props, rows)
        return out


def _collect(out, name, props, rows):
    names = [p[-1] for p in props]
    if name == "vertex":
        ix = [names.index(a) for a in "xyz"]
        out["points"] = np.asarray([[r[i] for i in ix] for r in rows], dtype=np.float64).reshape(-1, 3)
    elif name == "face":
        li = next(i for i, p in enumerate(props) if p[0] == "list")
        tris = [(r[li][0], r[li][j], r[li][j + 1]) for r in rows for j in range(1, len(r[li]) - 1)]
        out["faces"] = np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def read_cloud(path: str) -> np.ndarray:
    """points f64[N,3] of a `.ply` or a whitespace-separated `.xyz`"""
    if path.endswith(".ply"):
        return read_ply(path)["points"]
    return np.atleast_2d(np.loadtxt(path))[:, :3].astype(np.float64)


def _ns(args, key, default=None):
    return args.get(key, default) if isinstance(args, dict) else getattr(args, key, default)


@torch.no_grad()
def get_mectrics(args, gt, pred, gt_mesh=None) -> dict:
    """args.dataset in {snpp, arkit}, args.normalize; gt / pred: clouds [N,3]; gt_mesh: {"points", "faces"} (snpp)"""
    dev = "cuda"
    gt = torch.as_tensor(np.asarray(gt)).float().to(dev)
    pred = torch.as_tensor(np.asarray(pred)).float().to(dev)
    data = {"point_dist": None, "face_dist": None}  # (ARKit has no ground-truth mesh)
    if _ns(args, "dataset") == "snpp":
        assert gt_mesh is not None, "Ground truth mesh is required for SNPP dataset"
        verts = torch.as_tensor(gt_mesh["points"]).float().to(dev)
        faces = torch.as_tensor(gt_mesh["faces"]).long().to(dev)
        pd, fd = M.point_face_dist(pred, verts, faces, normalize=bool(_ns(args, "normalize")))
        data["point_dist"], data["face_dist"] = pd * MULTIPLIER, fd * MULTIPLIER
    if pred.ndim == 2:
        pred, gt = pred.unsqueeze(0), gt.unsqueeze(0)
    a, b = M.cd_unit_sphere(pred.contiguous(), gt.contiguous(), normalize=bool(_ns(args, "normalize")))
    data["cd_pred_gt"], data["cd_gt_pred"] = a * MULTIPLIER, b * MULTIPLIER
    return data


get_metrics = get_mectrics


def calculate_model_metrics(data: Dict, model_name: str, args) -> Dict:
    return {cfg: get_mectrics(args, data["faro"], pred, gt_mesh=data["faro_mesh"]) for cfg, pred in data["models"][model_name].items()}


def _scored(model_dir, args):
    path = os.path.join(model_dir, f"metrics{_ns(args, 'suffix', '')}.csv")
    if not os.path.exists(path):
        return set()
    with open(path, newline="") as f:
        return {r["model_config"] for r in csv.DictReader(f) if r.get("model_config")}


def _load_folder(root, args, faro_name, reduce_to_iphone):
    suffix = _ns(args, "suffix", "")
    scans, predictions = os.path.join(root, "scans"), os.path.join(root, f"predictions{suffix}")
    if not os.path.exists(predictions):
        return None
    iphone = read_cloud(os.path.join(scans, f"iphone{suffix}.ply"))
    data = {"iphone": iphone, "faro": None, "faro_mesh": None, "models": {}}
    for m in os.listdir(predictions):
        if m in ("iphone", "gt", "tsdf"):
            continue
        model = os.path.join(predictions, m)
        done = _scored(model, args)
        data["models"][model] = {}
        for fn in os.listdir(model):
            if not (fn.endswith(".ply") or fn.endswith(".xyz")) or fn[:-4] in done:
                continue
            pts = read_cloud(os.path.join(model, fn))
            if reduce_to_iphone:
                if iphone.shape[0] > pts.shape[0]:
                    continue  # (fewer points than the input scan: skipped, like the reference)
                if iphone.shape[0] < pts.shape[0]:
                    from .denoise import farthest_point_sampling

                    cloud = torch.as_tensor(pts).float().cuda()[None].contiguous()
                    pts = pts[farthest_point_sampling(cloud, iphone.shape[0])[1][0].cpu().numpy()]
            data["models"][model][fn[:-4]] = pts
    mesh = read_ply(os.path.join(scans, faro_name))
    data["faro"], data["faro_mesh"] = mesh["points"], (mesh if mesh["faces"] is not None else None)
    return data


def load_folder_snpp(root: str, args) -> Optional[Dict]:
    return _load_folder(root, args, "mesh_aligned_0.05.ply", reduce_to_iphone=True)


def load_folder_arkit(root: str, args) -> Optional[Dict]:
    return _load_folder(root, args, "faro.ply", reduce_to_iphone=False)


def handle_scene(scene_folder: str, args) -> None:
    data = (load_folder_snpp if _ns(args, "dataset") == "snpp" else load_folder_arkit)(scene_folder, args)
    if data is None:
        return
    for model in data["models"]:
        name = f"metrics{_ns(args, 'suffix', '')}.csv" + ("_normalized.csv" if _ns(args, "normalize") else "")
        # (the model key is an absolute path, so os.path.join returns it: <model dir>/<name>, the file _scored reads back)
        path = os.path.join(scene_folder, "metrics", model, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rows = []
        if os.path.exists(path):
            with open(path, newline="") as f:
                rows = list(csv.DictReader(f))
        have = {r["model_config"] for r in rows}
        for cfg, m in calculate_model_metrics(data, model, args).items():
            if cfg not in have:
                rows.append(dict(m, model_config=cfg))
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=COLUMNS)
            w.writeheader()
            for r in rows:
                w.writerow({k: ("" if r.get(k) is None else r.get(k)) for k in COLUMNS})
        torch.cuda.empty_cache()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_root", type=str, required=True)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--dataset", type=str, required=True, choices=["snpp", "arkit"])
    ap.add_argument("--single_dir", action="store_true")
    ap.add_argument("--normalize", action="store_true")
    ap.add_argument("--suffix", default="")
    args = ap.parse_args(argv)
    for f in os.listdir(args.data_root):
        handle_scene(os.path.join(args.data_root, f), args)


if __name__ == "__main__":
    main()
