"""object patch pipeline timings (kNN patches, FPS merge) at PU-Net object sizes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p2p_bridge_amd import denoise
for N, K in [(10000, 2048), (50000, 2048), (50000, 4096)]:
    g = torch.Generator().manual_seed(0)
    pcl = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=2).cuda()
    S = int(3 * N / K)
    def t(fn, n=3):
        fn(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3, r
    ms_seed, (seeds, _) = t(lambda: denoise.farthest_point_sampling(pcl, S))
    ms_knn, knn = t(lambda: denoise.knn_points(seeds, pcl, K=K, return_nn=True))
    merged = knn.knn.reshape(1, -1, 3).contiguous()
    ms_merge, _ = t(lambda: denoise.farthest_point_sampling(merged, N), n=1)
    print(f"N={N} K={K} S={S}: seed FPS {ms_seed:.2f} ms, kNN {ms_knn:.2f} ms, merge FPS {merged.shape[1]}->{N} {ms_merge:.1f} ms", flush=True)
