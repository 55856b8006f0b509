"""tests/golden/knn_sklearn.npz: an INDEPENDENT exact K-nearest-neighbour answer for the object pipeline's patch extraction.
The reference calls pytorch3d.ops.knn_points (denoise_object.py:91, dataloaders/punet.py:335), a pip dependency that is neither
under /root/reference nor installable here; its published contract (the K nearest by squared Euclidean distance, ascending) is
restated in oracle/cpu_ops.py knn_points. scikit-learn IS installed in the build container: its brute-force NearestNeighbors
(float64) answers the same question, so the restated contract gets an external anchor -- same neighbour SETS, same ascending
distances -- even though pytorch3d itself stays unpinned. Never run on the GPU box.     python tools/make_golden_knn.py"""
import os

import numpy as np
import sklearn
from sklearn.neighbors import NearestNeighbors

rng = np.random.default_rng(21)
n, s, K = 6000, 24, 512
u = rng.standard_normal((n, 3))
pts = (u / np.linalg.norm(u, axis=1, keepdims=True) * np.array([1.0, 0.7, 0.4]) + 0.01 * rng.standard_normal((n, 3))).astype(np.float32)
seeds = pts[rng.choice(n, s, replace=False)]
nn = NearestNeighbors(n_neighbors=K, algorithm="brute", metric="euclidean").fit(pts.astype(np.float64))
dist, idx = nn.kneighbors(seeds.astype(np.float64))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "knn_sklearn.npz")
np.savez_compressed(path, points=pts, seeds=seeds, K=np.array(K), idx=idx.astype(np.int32), dist2=(dist ** 2),
                    sklearn_version=np.array(sklearn.__version__))
print("wrote", path, os.path.getsize(path), "bytes; scikit-learn", sklearn.__version__)
