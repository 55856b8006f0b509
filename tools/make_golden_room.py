"""tests/golden/room_radius.npz: the radius lists of the room pipeline from scikit-learn ITSELF -- the call the reference makes
(denoise_room.py:454,464: `neighbors.KDTree(room_points, metric="l2").query_radius(center_points, r=0.3 | 0.5,
return_distance=False)`), run in the build container where scikit-learn is installed (it is a pip dependency of the reference,
unpinned in requirements.txt, absent from /root/reference; the fixture records the version that produced it). Never run on the
GPU box. The lists are stored ascending per centre (sklearn returns them in tree order; the pipeline only uses them as sets:
oracle/cpu_ops.py radius_query's contract), with the float64 squared distances of the pairs closest to the sphere so that a test
can tell a genuine difference from an fp32 / fp64 boundary tie.
    python tools/make_golden_room.py"""
import os

import numpy as np
import sklearn
from sklearn import neighbors

rng = np.random.default_rng(7)
# a synthetic "room": floor, two walls and clutter, 24000 points in a 6 x 4 x 2.6 m box (float32 coordinates, as the loaders give)
n = 24000
floor = np.c_[rng.uniform(0, 6, n // 3), rng.uniform(0, 4, n // 3), rng.normal(0, 0.004, n // 3)]
wall = np.c_[rng.uniform(0, 6, n // 6), rng.normal(0, 0.004, n // 6), rng.uniform(0, 2.6, n // 6)]
wall2 = np.c_[rng.normal(0, 0.004, n // 6), rng.uniform(0, 4, n // 6), rng.uniform(0, 2.6, n // 6)]
clutter = rng.uniform([0.5, 0.5, 0], [5.5, 3.5, 1.2], (n - n // 3 - 2 * (n // 6), 3))
points = np.concatenate([floor, wall, wall2, clutter]).astype(np.float32)
rng.shuffle(points)
centers = points[rng.choice(n, 48, replace=False)]  # (the pipeline's centres are points of the cloud: FPS picks)
out = {"points": points, "centers": centers, "sklearn_version": np.array(sklearn.__version__)}
tree = neighbors.KDTree(points, metric="l2")
for r in (0.3, 0.5):
    lists = tree.query_radius(centers, r=r, return_distance=False)
    flat = np.concatenate([np.sort(l) for l in lists]).astype(np.int32)
    off = np.zeros(len(lists) + 1, np.int64)
    off[1:] = np.cumsum([len(l) for l in lists])
    tag = f"r{int(r * 10):02d}"
    out[f"idx_{tag}"], out[f"off_{tag}"] = flat, off
    print(f"r = {r}: {off[-1]} pairs, {off[-1] / len(lists):.0f} per centre")
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "room_radius.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes; scikit-learn", sklearn.__version__)
