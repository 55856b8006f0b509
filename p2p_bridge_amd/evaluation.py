"""Output side of the object pipeline (the callers either side of the hot path): what `evaluate_objects.py`
and `models/evaluation.py` do around `patch_based_denoise` -- read noisy `.xyz` clouds, denoise, write `.xyz`, then
score every result against its clean cloud and mesh -- on this package's HIP metrics (`metrics.py`, csrc/chamfer.hip,
emd.hip, p2m.hip). Same function / class names, arguments and result keys as the reference:

  load_xyz / load_off            models/evaluation.py:253-279   (directories -> {name: tensor} / {name: {verts, faces}})
  write_array_to_xyz             utils/utils.py:5-10            (the file format the next stage reads back)
  input_iter                     evaluate_objects.py:48-67      (noisy cloud normalised to the unit sphere)
  get_metrics                    models/evaluation.py:206-246   (CD x 1000, EMD x 1000, model loss)
  evaluate, save_ptc             models/evaluation.py:64-203    (the in-training evaluation: sample the validation batches,
                                                                 score prediction / input / condition against the ground truth;
                                                                 the reference's matplotlib / wandb pictures are out of scope)
  calculate_cd, calculate_emd_exact_cuda   metrics/metrics.py:19-53, 111-136  (the `fast=False` branch)
  Evaluator / update_summary     models/evaluation.py:356-452   (cd_sph + p2f per shape, means, Summary_<dataset>.csv)
  denoise_and_evaluate           evaluate_objects.py:127-209 `sample` (resolutions x noise levels)

Differences, all at third-party edges that are not under /root/reference ("parity unpinned" where noted):
  * OFF meshes are parsed here (ASCII OFF, polygons fanned into triangles) instead of point_cloud_utils.load_mesh_vf;
  * `calculate_cd` (point_cloud_utils.chamfer_distance: mean nearest-neighbour EUCLIDEAN distance, both directions
    added) runs on the HIP Chamfer kernel + sqrt instead of a CPU KD-tree -- unpinned;
  * the `blensor` rotation uses a literal matrix instead of scipy's Rotation.from_euler("xyz", [-90, 0, 0]);
  * results are plain dicts / csv text (no pandas), same columns and `%.12f` format; logging is the caller's business.
"""
import csv
import os
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import metrics as M
from .punet_data import NormalizeUnitSphere

__all__ = ["load_xyz", "load_off", "write_array_to_xyz", "input_iter", "get_metrics", "calculate_cd",
           "calculate_emd_exact_cuda", "evaluate", "save_ptc", "Evaluator", "update_summary", "denoise_and_evaluate"]


# ------------------------------------------------------------------------------------------------ files


def load_xyz(xyz_dir: str) -> Dict[str, torch.Tensor]:
    """every `<name>.xyz` of a directory as f32[N, C] (whitespace-separated text, one point per line)"""
    out = {}
    for fn in sorted(os.listdir(xyz_dir)):
        if fn.endswith("xyz"):
            out[fn[:-4]] = torch.from_numpy(np.atleast_2d(np.loadtxt(os.path.join(xyz_dir, fn), dtype=np.float32)))
    return out


def _read_off(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """ASCII OFF -> (verts f32[V,3], triangles i64[F,3]); tolerates `OFF` fused with the counts line, comments and blank
    lines, and fans polygons with more than three corners"""
    with open(path) as f:
        tokens: List[str] = []
        for line in f:
            line = line.split("#", 1)[0].strip()
            if line:
                tokens.extend(line.split())
    if not tokens or not tokens[0].upper().startswith("OFF"):
        raise ValueError(f"{path}: not an OFF file")
    head = tokens[0][3:]
    tokens = ([head] if head else []) + tokens[1:]
    nv, nf = int(tokens[0]), int(tokens[1])
    pos = 3  # (vertices, faces, edges)
    verts = np.asarray(tokens[pos:pos + 3 * nv], dtype=np.float32).reshape(nv, 3)
    pos += 3 * nv
    tris = []
    for _ in range(nf):
        k = int(tokens[pos])
        corner = [int(t) for t in tokens[pos + 1:pos + 1 + k]]
        pos += 1 + k
        tris.extend((corner[0], corner[i], corner[i + 1]) for i in range(1, k - 1))
    return verts, np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def load_off(off_dir: str) -> Dict[str, Dict[str, torch.Tensor]]:
    out = {}
    for fn in sorted(os.listdir(off_dir)):
        if fn.endswith("off"):
            v, t = _read_off(os.path.join(off_dir, fn))
            out[fn[:-4]] = {"verts": torch.from_numpy(v), "faces": torch.from_numpy(t)}
    return out


def write_array_to_xyz(path: str, array) -> None:
    """`%8f` columns separated by one blank, rows by a newline, no trailing newline -- what np.loadtxt reads back"""
    array = np.asarray(array)
    row = " ".join(["%8f"] * array.shape[1])
    with open(path, "w") as f:
        f.write("\n".join(row % tuple(r) for r in array))


def input_iter(input_dir: str) -> Iterator[dict]:
    """noisy clouds of a directory, each moved / scaled into the unit sphere (centre and scale ride along)"""
    for fn in os.listdir(input_dir):
        if not fn.endswith("xyz"):
            continue
        pcl = torch.from_numpy(np.atleast_2d(np.loadtxt(os.path.join(input_dir, fn))).astype(np.float32))
        pcl, center, scale = NormalizeUnitSphere.normalize(pcl)
        yield {"pcl_noisy": pcl, "name": fn[:-4], "center": center, "scale": scale}


# ------------------------------------------------------------------------------------------------ metrics


def _points_last(*clouds):
    return tuple(c if c.shape[-1] == 3 else c.transpose(-1, -2) for c in clouds)


@torch.no_grad()
def calculate_cd(pred, gt) -> List[float]:
    """per cloud: mean_i min_j |p_i - g_j| + mean_j min_i |g_j - p_i| (Euclidean, not squared)"""
    assert pred.shape == gt.shape, f"CD calculation asserts same shape but pred shape: {pred.shape}, gt shape: {gt.shape}"
    pred, gt = _points_last(pred, gt)
    out = []
    for s in range(0, pred.shape[0], 4):
        d1, d2 = M.chamfer_dist_nograd(pred[s:s + 4].contiguous().float(), gt[s:s + 4].contiguous().float())
        out.append(d1.sqrt().mean(dim=1) + d2.sqrt().mean(dim=1))
    return torch.cat(out).cpu().tolist()


@torch.no_grad()
def calculate_emd_exact_cuda(pred, gt) -> List[float]:
    """auction assignment (eps 0.001, up to 10000 rounds): sqrt of the mean squared matched distance per cloud"""
    emd = M.emdModule()
    out = []
    for s in range(0, pred.shape[0], 4):
        dis, _ = emd(pred[s:s + 4].contiguous(), gt[s:s + 4].contiguous(), 0.001, 10000)
        out.extend(torch.mean(dis.detach(), dim=1).sqrt().cpu().tolist())
    return out


def get_metrics(gt, pred, model=None, fast: bool = True) -> Tuple[float, float, float]:
    """(Chamfer x 1000, EMD x 1000, mean model loss) of a batch of predictions; either tensor layout ([B,3,N] or
    [B,N,3]) is accepted and put points-first like the reference does"""
    if pred.shape[-1] < pred.shape[-2]:
        pred = pred.transpose(1, 2)
    if gt.shape[-1] < gt.shape[-2]:
        gt = gt.transpose(1, 2)
    if fast:
        p3, g3 = _points_last(pred, gt)
        cd = float(np.mean(M.calculate_cd_cuda(p3.contiguous(), g3.contiguous()))) * 1000
        loss = float(np.mean(model.loss(pred, gt).cpu().numpy())) if model is not None else 0
        # the reference averages the approximate EMD per chunk of four clouds first, then over the chunks
        chunks = [float(np.mean(M.earth_mover_distance_nograd(p, g, transpose=p.shape[-1] > p.shape[-2]).cpu().numpy()))
                  for p, g in zip(torch.split(pred, 4, dim=0), torch.split(gt, 4, dim=0))]
        return cd, float(np.mean(chunks)) * 1000, loss
    n = pred.shape[-1] - pred.shape[-1] % 128  # (a multiple of 128 points, as the reference's exact EMD wants)
    pred, gt = pred[..., :n].transpose(1, 2).contiguous(), gt[..., :n].transpose(1, 2).contiguous()
    cd = float(np.mean(calculate_cd(pred, gt))) * 1000
    loss = float(np.mean(model.loss(pred, gt).detach().cpu().numpy())) if model is not None else 0
    return cd, float(np.mean(calculate_emd_exact_cuda(pred, gt))) * 1000, loss


def save_ptc(name: str, ptc, out_dir: str, step: int) -> None:
    np.save("%s/%03d_%s.npy" % (out_dir, step, name), ptc.cpu().numpy())


def _cfg(cfg, *path, default=None):
    for key in path:
        if cfg is None:
            return default
        cfg = cfg.get(key) if isinstance(cfg, dict) else getattr(cfg, key, None)
    return default if cfg is None else cfg


@torch.no_grad()
def evaluate(model, val_loader, cfg, step: int, sampling: bool = False, save_npy: bool = False, fast: bool = False) -> dict:
    """sample the first `cfg.sampling.accum_iter` validation batches and score them: {"cd", "emd", "mse"} of the
    prediction, "*_noisy" of the sampler's input and "*_cond" of the condition cloud (PVDCond models) against the ground
    truth, over the largest multiple of 128 points. save_npy (with sampling): `<cfg.out_sampling>/<step>_{pred,noisy,gt,cond}.npy`"""
    from .train import get_data_batch

    device = next(model.parameters()).device
    preds, starts, gts, conds = [], [], [], []
    for i, batch in enumerate(val_loader):
        data = get_data_batch(batch=batch, cfg=cfg)
        x_gt, x_cond, x_start = (None if t is None else t.to(device) for t in (data["x_gt"], data["x_cond"], data["x_start"]))
        out = model.sample(x_start=x_start, x_cond=x_cond, clip=bool(_cfg(cfg, "diffusion", "clip", default=False)),
                           use_ema=bool(_cfg(cfg, "use_ema", default=False)), verbose=False)
        preds.append(out["x_pred"]), starts.append(out["x_start"]), gts.append(x_gt)
        if _cfg(cfg, "model", "type") == "PVDCond" and x_cond is not None:
            conds.append(x_cond[:, :3, :])
        if i >= int(_cfg(cfg, "sampling", "accum_iter", default=1)) - 1:
            break
    pred, x_gt = torch.cat(preds), torch.cat(gts)
    x_start = torch.cat(starts) if starts and starts[0] is not None else None
    x_cond = torch.cat(conds) if conds else None
    n = pred.shape[-1] - pred.shape[-1] % 128
    pred, x_gt = pred[..., :n], x_gt[..., :n]
    metrics = dict(zip(("cd", "emd", "mse"), get_metrics(x_gt, pred, model=model, fast=fast)))
    for tag, cloud in (("noisy", x_start), ("cond", x_cond)):
        if cloud is not None:
            metrics.update(zip((f"cd_{tag}", f"emd_{tag}", f"mse_{tag}"), get_metrics(x_gt, cloud[..., :n], model=model, fast=fast)))
    if sampling and save_npy:
        out_dir = _cfg(cfg, "out_sampling")
        for name, cloud in (("pred", pred), ("noisy", x_start), ("gt", x_gt), ("cond", x_cond)):
            if cloud is not None:
                save_ptc(name, cloud[..., :n], out_dir, step)
    return metrics


# ------------------------------------------------------------------------------------------------ per-shape evaluation

_BLENSOR_ROT = ((1.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, -1.0, 0.0))  # rotation by -90 degrees about x


class Evaluator:
    """scores the `.xyz` results of one run against `<dataset_root>/<dataset>/pointclouds/test/<res_gts>/` and
    `.../meshes/test/`: Chamfer on the unit sphere and the two-sided point-to-mesh distance per shape"""

    def __init__(self, output_pcl_dir, dataset_root, dataset, summary_dir, experiment_name, device="cuda",
                 res_gts="8192_poisson"):
        self.output_pcl_dir, self.dataset_root, self.dataset = output_pcl_dir, dataset_root, dataset
        self.summary_dir, self.experiment_name, self.device, self.res_gts = summary_dir, experiment_name, device, res_gts
        self.gts_pcl_dir = os.path.join(dataset_root, dataset, "pointclouds", "test", res_gts)
        self.gts_mesh_dir = os.path.join(dataset_root, dataset, "meshes", "test")
        self.load_data()

    def load_data(self):
        self.pcls_up = load_xyz(self.output_pcl_dir)
        self.pcls_high = load_xyz(self.gts_pcl_dir)
        self.meshes = load_off(self.gts_mesh_dir)
        self.pcls_name = list(self.pcls_up.keys())

    def run(self) -> Dict[str, Dict[str, float]]:
        results = {}
        for name in self.pcls_name:
            up = self.pcls_up[name]
            if up.dim() != 2 or name not in self.pcls_high:
                continue  # (malformed result / shape without ground truth: skipped, like the reference)
            up = up[:, :3].unsqueeze(0).to(self.device)
            high = self.pcls_high[name].unsqueeze(0).to(self.device)
            verts, faces = self.meshes[name]["verts"].to(self.device), self.meshes[name]["faces"].to(self.device)
            cd_sph = M.chamfer_distance_unit_sphere(up, high)[0].item()
            cloud = up[0]
            if "blensor" in self.experiment_name:  # those scans are stored rotated against their meshes
                cloud = cloud.matmul(torch.tensor(_BLENSOR_ROT, device=cloud.device).t())
            p2f = M.point_mesh_bidir_distance_single_unit_sphere(pcl=cloud, verts=verts, faces=faces).item()
            results[name] = {"cd_sph": cd_sph, "p2f": p2f}
        self.results = results
        self.means = {k: float(np.mean([r[k] for r in results.values()])) for k in ("cd_sph", "p2f")} if results else {}
        if results:
            update_summary(os.path.join(self.summary_dir, "Summary_%s.csv" % self.dataset), model=self.experiment_name,
                           metrics={"cd_sph(mean)": self.means["cd_sph"], "p2f(mean)": self.means["p2f"]})
        return results


def update_summary(path: str, model: str, metrics: Dict[str, float]) -> Dict[str, Dict[str, str]]:
    """one row per model, one column per metric, `%.12f`; an existing file keeps its other rows and columns"""
    table: Dict[str, Dict[str, str]] = {}
    columns: List[str] = []
    if os.path.exists(path):
        with open(path, newline="") as f:
            rows = list(csv.reader(f))
        if rows:
            columns = [c.strip() for c in rows[0][1:]]
            for r in rows[1:]:
                if r:
                    table[r[0].strip()] = {c: v.strip() for c, v in zip(columns, r[1:])}
    for c in metrics:
        if c not in columns:
            columns.append(c)
    table.setdefault(model, {}).update({c: "%.12f" % v for c, v in metrics.items()})
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([""] + columns)
        for name, row in table.items():
            w.writerow([name] + [row.get(c, "") for c in columns])
    return table


# ------------------------------------------------------------------------------------------------ the script's loop


@torch.no_grad()
def denoise_and_evaluate(model, cfg, data_path: str, dataset_root: str, output_root: str, dataset: str = "PUNet",
                         resolutions=(10000, 50000), noises=(0.01, 0.02, 0.03), save_title: str = "P2P-Bridge",
                         patch_size: int = 2048, evaluate: bool = True) -> Dict[str, dict]:
    """for every (resolution, noise level): denoise the clouds of `<data_path>/<dataset>_<res>_poisson_<noise>/` patch by
    patch, write them under `<output_root>/<dataset>/<title>_<res>_<noise>/pcl/`, score them with Evaluator.
    cfg: the keys `patch_based_denoise` reads (use_ema, steps, k, save_intermediate). Returns {run: per-shape results}."""
    from .denoise import _get, patch_based_denoise

    device = next(model.parameters()).device
    if _get(cfg, "use_ema"):
        save_title += "_ema"
    save_title += f"_steps_{_get(cfg, 'steps')}"
    out_root = os.path.join(output_root, dataset)
    summary = {}
    for res in resolutions:
        for noise in noises:
            input_dir = os.path.join(data_path, "%s_%s_poisson_%s" % (dataset, res, noise))
            output_dir = os.path.join(out_root, f"{save_title}_{res}_{noise}")
            os.makedirs(os.path.join(output_dir, "pcl"), exist_ok=True)
            for data in input_iter(input_dir):
                model.eval()
                denoised, steps = patch_based_denoise(model=model, pcl_noisy=data["pcl_noisy"].to(device),
                                                      patch_size=patch_size, seed_k=_get(cfg, "k") or 3, cfg=cfg,
                                                      save_intermediate=bool(_get(cfg, "save_intermediate")))
                denoised = denoised.cpu() * data["scale"] + data["center"]
                write_array_to_xyz(os.path.join(output_dir, "pcl", data["name"] + ".xyz"), denoised.numpy())
                if steps is not None:
                    for i, item in enumerate(steps):
                        path = os.path.join(output_dir, "steps", data["name"], data["name"] + f"_{i}.xyz")
                        os.makedirs(os.path.dirname(path), exist_ok=True)
                        write_array_to_xyz(path, (item.cpu() * data["scale"] + data["center"]).numpy())
            if evaluate:
                summary[f"{res}_{noise}"] = Evaluator(output_pcl_dir=os.path.join(output_dir, "pcl"),
                                                      dataset_root=dataset_root, dataset=dataset, summary_dir=output_dir,
                                                      experiment_name=save_title, device=str(device),
                                                      res_gts=f"{res}_poisson").run()
    return summary
