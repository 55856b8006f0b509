#!/usr/bin/env python
"""bench.py -- denoised points/sec of the P2P-Bridge sampler on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: `P2PB.sample` of B=32 synthetic PU-Net-shaped
patches of 8192 points through T=30 bridge steps of the PVDS_PUNet network (BASELINE config 2,
data.npoints=8192, seeded random weights -- no checkpoints/datasets offline). Inputs are resident in HBM
before the timed region. With --gpus N (launched by torch.distributed.run) every rank denoises its own
B patches: patch-level sharding, no data-path collective, weak scaling; time = max over ranks.

Prints ONE JSON line (rank 0) with the driver's fields + "roofline" (dominant kernel, live HIP-event
timing) + "cpu_baseline" (the CPU oracle timed on this host on a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # CPU-baseline leg: two OpenMP runtimes (torch, oracle) must not spin

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PVDS = dict(
    data=dict(npoints=8192),
    diffusion=dict(timesteps=1000, sampling_timesteps=30, objective="pred_noise", sampling_strategy="DDPM",
                   loss_type="mse", beta_start=1e-4, beta_end=0.02, t0=1e-4, T=1.0, ot_ode=True),
    model=dict(type="PVD", ema=False, in_dim=3, extra_feature_channels=0, out_dim=3, time_embed_dim=64, dropout=0.15,
               PVD=dict(use_global_embedding=True, global_embedding_dim=1024, feat_embed_dim=32,
                        attention_type="linear", attention_heads=4, attentions=[0, 0, 0, 1],
                        channels=[32, 64, 128, 256, 512], voxel_resolutions=[32, 16, 8, 8], n_sa_blocks=[1, 2, 1, 1],
                        n_fp_blocks=[1, 2, 1, 1], radius=[0.1, 0.2, 0.4, 0.8], out_mlp=128)))

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
BF16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz)
# (v_mfma_f32_32x32x16_f16 runs at the same rate.) A split kernel spends three (f16x3, the default) or six (bf16x6)
# 16-bit MFMA products per fp32 product: its matrix-pipe ceiling in fp32-equivalent FLOP/s
SPLIT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6


def split_products():
    from p2p_bridge_amd import fused

    return 3 if fused.conv_math() == "f16x3" else 6


def split_peak_tflops():
    return BF16_MFMA_PEAK_TFLOPS / split_products()


def split_peak_basis():
    return (f"dense 16-bit MFMA peak 2516.6 / {split_products()} products per fp32 product "
            + ("(f16x3: fp16-pair split operands, fp32 accumulate)" if split_products() == 3
               else "(bf16x6 split operands, fp32 accumulate)"))
CONV_ALGO_FLOP_PER_SAMPLE_EVAL = 43.88e9  # Conv3d FLOPs per sample per network evaluation (PVDS)


def conv_roofline(model, x_start, reps=10):
    """Second kernel = the 3x3x3 voxel convolution AS THE SAMPLER RUNS IT: conv3d_k3_compact_kernel on the widest
    r = 16 instance (fp_layers.2.1, second convolution, C 128 -> 128: far-field form, folded AdaGN + Swish operand
    transform, GroupNorm statistics epilogue). The kernel computes only the outputs within two voxels of an occupied
    voxel (set D2, csrc/conv3d.hip) and writes analytic constants elsewhere, so its work depends on the occupancy of the
    input: the launch is captured from ONE real network evaluation of the bench's own patches (same tensors, same
    lists), then re-issued `reps` times between HIP events on the stream it is launched on.
    `achieved` counts the ALGORITHMIC FLOPs of what the layer must produce with this formulation: 2*27*Cin*Cout per LISTED
    output voxel (the exact-constant voxels cost no matrix work by construction); `dense_equivalent` is the same
    launch priced as the dense convolution the reference runs (all 16^3 voxels)."""
    import glob

    from p2p_bridge_amd import fused

    pv = model.model.fp_layers[2][1]
    target = pv.voxel_layers[4]
    captured = {}
    orig = fused.conv3d_k3_compact

    def spy(x, conv, lists, counts, which, *a, **k):
        if conv is target and "args" not in captured:
            captured["args"] = (x, conv, lists, counts, which) + a
            captured["kw"] = k
        return orig(x, conv, lists, counts, which, *a, **k)

    fused.conv3d_k3_compact = spy
    try:
        model.eval()
        with torch.no_grad():
            model.model(x_start, torch.full((x_start.shape[0],), 500.0, device=x_start.device))
        model.train()
    finally:
        fused.conv3d_k3_compact = orig
    if "args" not in captured:
        return None
    args, kw = captured["args"], captured["kw"]
    x, conv, lists, counts, which = args[:5]
    B, r, C = x.shape[0], x.shape[1], x.shape[4]
    listed = int(counts[which].sum().item())
    flops = 2.0 * 27 * conv.in_channels * conv.out_channels * listed
    dense = 2.0 * 27 * conv.in_channels * conv.out_channels * B * r ** 3
    with torch.no_grad():
        for _ in range(3):
            orig(*args, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            orig(*args, **kw)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM-side traffic of this launch from the committed PMC passes (tools/pmc_conv_instances.sh: the same launch, B = 32), the
    # counters read as for the first kernel: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch
    traffic, basis = None, None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv_compact.csv")), reverse=True)
    if cands and B == 32 and r == 16 and C == 128:
        vals = {}
        for line in open(cands[0]).read().splitlines()[1:]:
            k, v = line.split(",")[:2]
            vals[k] = float(v)
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            traffic = round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
            algo = 4 * B * r ** 3 * C + 4 * listed * conv.out_channels + 4 * 27 * conv.in_channels * conv.out_channels
            basis = (f"profiles/{os.path.basename(cands[0])}: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch; algorithmic = the pre-split operand "
                     f"grid once + the listed outputs + the weight pack = {algo} B (a brick's 6x10x10 halo overlaps its neighbours' 2.34 x: "
                     "the re-reads are served by the XCD's L2 as far as the brick order keeps them there)")
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": round(split_peak_tflops(), 1), "unit": "TFLOP/s",
            "frac": round(achieved / split_peak_tflops(), 4), "traffic": traffic, "traffic_basis": basis,
            "kernel": f"conv3d_k3_compact_kernel<{r},XF> C{conv.in_channels}->{conv.out_channels} r{r} B{B} "
                      f"(fp_layers.2.1.voxel_layers.4, the launch the sampler issues)",
            "listed_output_voxels": listed, "grid_voxels": B * r ** 3,
            "listed_fraction": round(listed / float(B * r ** 3), 4),
            "dense_equivalent_tflops": round(dense / (ms * 1e-3) / 1e12, 2),
            "peak_basis": split_peak_basis(),
            "flop_per_launch": flops, "ms_per_launch": round(ms, 4)}


def gemm_roofline(model, B, P, reps=10):
    """Dominant kernel of the sampler (largest share of the critical stream in profiles/r0N*_per_eval.csv): the global
    embedding's last layer (Pnet2Stage mlp2, 512 -> 1024 channels over all P points of every patch;
    models/pvcnn.py:905-932) -- pw_pp512_kernel<XF=true, POOL=true> (csrc/pw_pp512.h: the ping-pong GEMM on 512-channel x
    128-position tiles; pw_split_kernel under P2PB_EXPERIMENT pw_pp=0 or for layers without whole 512-channel blocks): split-operand GEMM in the arithmetic fused.conv_math()
    selects (f16x3 by default) that applies the previous layer's folded GroupNorm + Swish to its operand on load and whose
    epilogue emits the GroupNorm statistics and the per-channel {min, max} the max-pool is formed from -- the
    1024-channel output is never written. Timed live with HIP events on torch's current stream, launched exactly as
    the sampler launches it, right after the timed sampler runs (the launch time follows the clock the part holds: 0.82-1.02 ms
    for the same binary and operands depending on what ran before -- profiles/r05c_gemm_operand_and_batch.txt; behind the
    sampler it sits in the middle of that range). `achieved` = algorithmic fp32 FLOPs (2*Cin*Cout per position) / mean launch time."""
    from p2p_bridge_amd import fused

    conv = model.model.global_pnet.mlp2.shared_mlp_1.mlp[0]
    ci, co = conv.in_channels, conv.out_channels
    x = torch.randn(B, ci, P, device="cuda")
    flops = 2.0 * B * P * ci * co
    with torch.no_grad():
        # as in the sampler: the previous layer's folded GroupNorm + Swish applied to the operand on load
        sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda")
        for _ in range(3):
            fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fused.pw_conv(x, conv, sc, sh, swish=True, pool_u=0, store=False)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    achieved = flops / (ms * 1e-3) / 1e12
    split = fused.use_split_pw(ci, co, P, None)
    peak = split_peak_tflops() if split else F32_MFMA_PEAK_TFLOPS
    traffic = None
    try:
        vals = {}
        pingpong = split and fused.conv_math() == "f16x3" and _pw_pp_on() and co % 512 == 0 and ci % 64 == 0 and B * ((P + 127) // 128) * (co // 256) >= 1024
        import glob

        cands = (sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_pw_pp512_512_1024_pool.csv")), reverse=True) if pingpong else [])
        if not pingpong:
            cands = [os.path.join(ROOT, "profiles", f"{t}_pmc_pw_split_512_1024_pool.csv") for t in ("r02f", "r02", "r01")]
        pmc = next(q for q in cands if os.path.exists(q))  # (newest round's PMC passes of the kernel this launch runs)
        for line in open(pmc):
            k, v = line.split(",")[:2]
            if k in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[k] = float(v)
        traffic = round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0) if split else None
    except (OSError, KeyError, ValueError, StopIteration):
        pass
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_basis": f"profiles/{os.path.basename(pmc) if traffic is not None else 'r0N_pmc_pw_*_512_1024_pool.csv'}: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch "
                             "(FETCH_SIZE counts half of 8- and 16-byte-per-lane streaming reads on gfx950: calibrated on a "
                             f"1 GiB read, tools/pmc_calib.sh); algorithmic input + weights = {4 * B * P * ci + 4 * ci * co} B: "
                             "the activation tile is staged once per 512-channel block (2 x for this layer), the blocks of one "
                             "tile run side by side on one XCD and share it in that XCD's L2",
            "kernel": (f"pw_pp512_kernel<XF=true,POOL=true> {ci}->{co} P{P} B{B} (global_pnet.mlp2.shared_mlp_1)" if pingpong
                       else f"pw_split_kernel<XF=true,POOL=true,WM=4> {ci}->{co} P{P} B{B} (global_pnet.mlp2.shared_mlp_1)"),
            "peak_basis": split_peak_basis() if split else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
            "frac_of_six_product_ceiling": round(achieved / SPLIT_PEAK_TFLOPS, 4) if split else None,
            "flop_per_launch": flops, "ms_per_launch": round(ms, 4)}


HBM_PEAK_TBPS = 8.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# algorithmic bytes of the custom (index / gather / scatter) ops, MB per sample per network evaluation, PVDS @ N = 8192
HBM_GROUPS = [
    ("voxelize (a9-a10: coordinates, counting sort, gather into the voxel-major / pre-split grid)", 28.43,
     ("voxel_coords_kernel", "vox_count_kernel", "vox_scan_kernel", "vox_fill_kernel", "vox_sort_kernel", "vox_gather_cl", "transpose_cn_kernel"),
     "8 small launches per level on 16 clouds: the coordinate half (voxel_coords, count / scan / fill / sort: one workgroup per cloud "
     "for the deterministic reductions and the scan) is latency-bound and runs on the geometry stream beside the dense layers; the "
     "gathers move 60 % of the bytes at ~2 TB/s"),
    ("devoxelize (a12)", 26.11, ("devox_cl",),  # (devox_cl_kernel and, for C % 64 == 0, devox_cl4_kernel)
     "8-corner gather of voxel-major rows (256 B per corner and point at C = 64): L2-resident grid, the point-major -> channel-major "
     "transpose through LDS"),
    ("grouping (a17: group_sub / group_stats, the set abstraction's gathered first layer)", 19.86, ("group_sub_kernel", "group_stats_kernel"),
     "row gathers of the ungrouped tensor (L2-resident: 1 MB per sample); since round 5 the statistics pass (group_stats) writes "
     "nothing and the consuming GEMM gathers its own operand, so most of the reference op's bytes are never moved -- a fraction "
     "near or above 1 here means avoided traffic, not bandwidth"),
    ("3-NN interpolation (a19: cell grid build, search, interpolate + add)", 12.49,
     ("three_nn_kernel", "three_nn_cells_kernel", "nn_cells_build_kernel", "three_interp_add_kernel"),
     "search-bound, not byte-bound: 53.7 M pair distances per sample are the work (cell-grid search for m >= 512 centres, brute force "
     "below); the bytes are what the reference's O(N M) scan would stream"),
    ("ball query (a16)", 0.51, ("ball_query",), "compute-bound distance tests from LDS-resident clouds; 0.5 MB of output per sample"),
    ("FPS (a14)", 0.14, ("fps_kernel", "fps_grid", "fps_coop"),
     "a dependent chain of M - 1 argmax rounds per cloud (one workgroup each): latency-bound by construction, hidden on the geometry stream"),
    ("gather (a15)", 0.08, ("gather_kernel",), "three coordinates per centre: launch-bound"),
]

# The compute-side yardstick of the three search kernels (bytes are the wrong yardstick for L2-resident,
# latency-bound geometry): the pair-distance evaluations of the REFERENCE's brute-force formulation per sample and evaluation at
# PVDS / 8192 points (levels 8192 -> 2048 -> 512 -> 128 -> 32), against the fp32 vector pipe: 256 CUs x 4 SIMDs x 16 lanes x
# 2.4 GHz = 39.3 T lane-instructions/s (MI355X_MICROARCH.md: 157.3 TFLOP/s = 2 FLOP x 2 (packed) x that), 8 instructions per
# pair (3 subtractions, 1 multiply, 2 fused multiply-adds, compare, select) = 4.9 T pairs/s for the whole chip.
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9
PAIR_PEAK_PER_S = VALU_LANE_OPS_PER_S / 8.0
_LEVELS = [(8192, 2048), (2048, 512), (512, 128), (128, 32)]
PAIR_GROUPS = {
    # group prefix -> (pair evaluations per sample and evaluation, CUs a launch can occupy at `b` clouds, what the number means)
    "FPS (a14)": (sum((m - 1) * n for n, m in _LEVELS), lambda b: b,
                  "(M - 1) N running-distance updates per level, EVERY one performed (nothing to prune below 16384 points); one "
                  "workgroup = one CU per cloud by construction, so frac_on_occupied_cus is the kernel's VALU efficiency"),
    "ball query (a16)": (sum(m * n for n, m in _LEVELS), lambda b: 256,
                         "M N distance tests of the reference's scan; the kernel stops a centre at its 32nd hit, so the rate "
                         "counts tests it did not need to make"),
    "3-NN interpolation (a19": (53.7e6, lambda b: 256,
                                "N M distances of the reference's scan; the cell-grid search visits ~76 candidates per point "
                                "instead of up to 2048: an EFFECTIVE rate (work avoided counts as work done)"),
}


def hbm_kernels(patches_per_chain):
    """The HBM side of the roofline (SURVEY 8d 'per-kernel HBM fractions from rocprof alongside'): per custom-op group the
    algorithmic bytes of one chain-evaluation (SURVEY's per-sample figure x the patches a sampler chain evaluates), the kernel
    time per evaluation from the newest committed rocprofv3 kernel trace of this bench (profiles/r*_per_eval.csv, produced by
    tools/profile_round.sh from the same command), and the fraction of the 8 TB/s HBM peak. Not measured live: it is a table
    of profile data, the file it came from is named."""
    import glob

    import re

    # the newest set BY NAME (rNN + letter: r05c > r05b > r04h): file times mean nothing in a fresh checkout -- the driver's box and
    # every gpurun snapshot -- where sorting by mtime picked an arbitrary round-1 table (found in profiles/r05c_bench_line.json)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_per_eval.csv"))
                   if re.fullmatch(r"r\d\d[a-z]?_per_eval\.csv", os.path.basename(f)))
    if not files:
        return None
    src = files[-1]
    rows = []
    for line in open(src):
        if line.startswith("#") or line.startswith("pct,"):
            continue
        parts = line.rstrip("\n").split(",", 4)
        if len(parts) == 5:
            rows.append((float(parts[3]), parts[4].strip('"')))
    out = []
    for name, mb, keys, why in HBM_GROUPS:
        ms = sum(t for t, k in rows if any(key in k for key in keys))
        if ms <= 0:
            continue
        nbytes = mb * 1e6 * patches_per_chain
        tbps = nbytes / (ms * 1e-3) / 1e12
        row = {"group": name, "algorithmic_bytes": int(nbytes), "ms_per_eval": round(ms, 4), "TB_per_s": round(tbps, 3),
               "frac_of_8TBps": round(tbps / HBM_PEAK_TBPS, 4), "note": why}
        for prefix, (pairs, cus, what) in PAIR_GROUPS.items():
            if name.startswith(prefix):
                rate = pairs * patches_per_chain / (ms * 1e-3)
                occ = min(256, cus(patches_per_chain))
                row.update({"pair_evaluations": int(pairs * patches_per_chain), "Gpairs_per_s": round(rate / 1e9, 1),
                            "frac_of_valu_pair_peak": round(rate / PAIR_PEAK_PER_S, 4), "cus_occupied": occ,
                            "frac_on_occupied_cus": round(rate / (PAIR_PEAK_PER_S * occ / 256.0), 4),
                            "compute_yardstick": what + f"; peak = {PAIR_PEAK_PER_S / 1e12:.2f} T pairs/s (39.3 T fp32 lane-"
                                                        "instructions/s / 8 per pair)"})
        out.append(row)
    return {"source": f"profiles/{os.path.basename(src)} (rocprofv3 --kernel-trace of bench.py, one {patches_per_chain}-patch chain-evaluation)",
            "kernels": out}


def conv_math_note():
    from p2p_bridge_amd import fused

    m = fused.conv_math()
    if m == "f16x3":
        return ("fp32 operands, fp32 accumulate, fp32 results; products through the 16-bit matrix pipe as f16x3: every "
                "operand an fp16 pair of its scaled value (22 significand bits), 3 exact MFMA products per fp32 product, "
                "<= 3 * 2^-22 relative; error vs fp64 at or below the exact-fp32 MFMA kernel's (tests/test_conv_math_gpu.py)")
    if m == "bf16x6":
        return ("fp32 results; products as bf16x6 split operands (3 bf16 terms each, 6 MFMA products, fp32 accumulate; "
                "error vs fp64 equal to the fp32 MFMA kernel's)")
    return "exact-fp32 MFMA"


def _pw_pp_on():
    from p2p_bridge_amd import _experiment

    return _experiment.get("pw_pp", "1") != "0"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_worker(sd_path, n_points, T, patches, budget_s, threads, cpu0):
    """one process of the CPU baseline (`bench.py --cpu-worker ...`): `threads` threads pinned to the logical CPUs cpu0 .. cpu0 +
    threads - 1, its own `patches` patches through the oracle's sampler; prints one JSON line {dt_step, steps, one}"""
    try:
        os.sched_setaffinity(0, set(range(cpu0, cpu0 + threads)))
    except (AttributeError, OSError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.set_num_threads(threads)
    from oracle import cpu_ops, net_ref

    cpu_ops.set_threads(threads)
    import copy

    sd = torch.load(sd_path)
    cfg = copy.deepcopy(PVDS)
    cfg["data"]["npoints"] = n_points
    x, _ = net_ref.synthetic_patches(patches, n_points, seed=cpu0)
    net = net_ref.RefNet(cfg, sd, vox_mode="tree")
    t0 = time.perf_counter()
    net_ref.sample(net, cfg, x, steps=1, log_count=1)  # warm-up, and the estimate of one step
    one = time.perf_counter() - t0
    steps = max(2, min(T, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    net_ref.sample(net, cfg, x, steps=steps, log_count=1)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"dt_step": dt, "steps": steps, "one": one}), flush=True)


def cpu_baseline(sd, n_points, T, patches=2, budget_s=20.0):
    """The CPU oracle (C ops + torch CPU dense layers) on a bounded sample of the same workload, USING THE HOST: P worker
    processes x 16 threads, each pinned to its own 16 logical CPUs and denoising its own `patches` patches (patches are independent:
    the same patch-parallelism the GPU ranks use), P = physical cores / 16 (review r4 item 8: one 16-thread process left 7/8 of the
    host idle, and more threads in ONE process only contend -- profiles/r03b_cpu_baseline_threads.txt). Every worker runs the FULL
    T-step sampler when that fits the time budget, else as many bridge steps as fit, extrapolated (every step costs the same: one
    network evaluation + an elementwise update). value = sum over workers of patches x points / (T x its seconds per step), all
    workers running at the same time. P2PB_CPU_THREADS / P2PB_CPU_PROCS override."""
    import subprocess
    import tempfile

    host = os.cpu_count() or 1
    threads = min(host, int(os.environ.get("P2PB_CPU_THREADS", "16")))
    physical = max(1, host // 2)  # (2 hardware threads per core on the GPU boxes' EPYCs; the first `physical` logical CPUs are distinct cores)
    procs = int(os.environ.get("P2PB_CPU_PROCS", str(max(1, physical // threads))))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sd.pt")
        torch.save({k: v.detach().cpu() for k, v in sd.items()}, path)
        cmd = lambda i: [sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(n_points), str(T), str(patches),
                         str(budget_s), str(threads), str(i * threads)]
        t0 = time.perf_counter()
        ps = [subprocess.Popen(cmd(i), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
        outs = [p.communicate()[0] for p in ps]
        wall = time.perf_counter() - t0
    rs = [json.loads(o.strip().splitlines()[-1]) for o in outs if o.strip()]
    if not rs:
        raise RuntimeError("cpu_baseline: no worker finished")
    value = sum(patches * n_points / (r["dt_step"] * T) for r in rs)
    steps = min(r["steps"] for r in rs)
    dt = max(r["dt_step"] for r in rs)
    return {"value": round(value, 2), "unit": "points/s", "cores": threads * len(rs), "processes": len(rs), "threads_per_process": threads,
            "host_cores": host, "host_cpu": cpu_model(), "kind": "port",
            "sample": f"{len(rs)} processes x {patches} patches x {n_points} pts, {steps} of T={T} bridge steps timed after 1 warm-up step "
                      f"in every process, all at once (slowest {dt:.2f} s/step; {wall:.1f} s of wall time in all)"
                      + ("" if steps == T else f", extrapolated to T={T}")}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":  # (a process of cpu_baseline)
        a = sys.argv[2:]
        return cpu_worker(a[0], int(a[1]), int(a[2]), int(a[3]), float(a[4]), int(a[5]), int(a[6]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the bf16x6 comparison leg (profiling runs)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the config-3 training-step leg")
    ap.add_argument("--no-pvdl", action="store_true", help="skip the BASELINE config-4 leg (full-width PVDL, 8 x 50000 points)")
    ap.add_argument("--backend", default="nccl", help="process-group backend: nccl (= RCCL over xGMI); gloo only with --dry-run")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / timing protocol only, no GPU work (CPU self-test of the N-rank path)")
    args = ap.parse_args()
    from p2p_bridge_amd import sharding

    if args.backend != "nccl" and not args.dry_run:
        raise SystemExit("--backend other than nccl is for --dry-run only: the product has no CPU path")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    mode, world = sharding.launch_plan(args.gpus, os.environ, ndev if args.backend == "nccl" else args.gpus, args.backend)
    if mode == "spawn":  # `python bench.py --gpus N`: become N ranks, one per GPU (reference: train.py:20-46,229)
        raise SystemExit(sharding.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], world))
    if not args.dry_run and ndev == 0:
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    rank = local_rank = 0
    dist = None
    if mode == "rank":
        import torch.distributed as dist

        rank, local_rank, world = sharding.init_rank(args.backend)
    elif ndev:
        torch.cuda.set_device(0)
    if args.dry_run:
        return dry_run(args, dist, rank, world)

    from p2p_bridge_amd.synthetic import synthetic_patches
    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.pvcnn_unet import PVCNN2Unet

    import copy

    cfg = copy.deepcopy(PVDS)
    cfg["data"]["npoints"] = args.points
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in PVCNN2Unet(cfg).state_dict().items()}
    model = product.build_model(cfg, sd, device=f"cuda:{local_rank}")
    x_start, _ = synthetic_patches(args.batch, args.points, seed=rank)  # every rank denoises its OWN patches
    x_start = x_start.cuda()

    def one():
        return model.sample(x_start=x_start, steps=args.T, log_count=1, verbose=False, graph=bool(args.graph))

    for _ in range(args.warmup):
        out = one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0  # this rank's own work (reported per rank; NOT what `value` is computed from)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt_local, device="cuda")
    assert torch.isfinite(out["x_pred"]).all()

    res = result_line(args, world, dt, dist)
    # every rank's own time / throughput and the device it ran on, gathered over the process group (collective)
    res["ranks"] = sharding.rank_evidence(dt_own, args.batch * args.points * args.steps, local_rank)
    if world > 1:
        assert res["ranks"]["distinct_devices"] == world, res["ranks"]
    res["config"]["conv_math"] = conv_math_note()
    if rank == 0:
        res["roofline"] = gemm_roofline(model, args.batch, args.points)
        res["roofline"]["second_kernel"] = conv_roofline(model, x_start)  # the voxel convolution the sampler runs
        evals = args.T
        res["roofline"]["sampler_dense_tflops"] = round(
            61.35e9 * args.batch * evals / (dt / args.steps) / 1e12, 2)  # 61.35 GFLOP/sample/eval
        chains = model._sampler_chains(x_start)
        res["roofline"]["hbm_kernels"] = hbm_kernels(args.batch // chains)
        if world == 1 and "P2PB_CONV_MATH" not in os.environ and not args.no_alt_math:
            res["alt_math"] = alt_math_leg(cfg, sd, x_start, args)
        if world == 1 and not args.no_train_step:
            del model  # (free the sampler's graph pool first)
            res["train_step"] = train_step_leg()
        if world == 1 and not args.no_pvdl:
            res["pvdl"] = pvdl_leg()
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, args.points, args.T)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_step_leg(steps=8, warmup=4, B=8, N=2048):
    """NOT `value`: BASELINE config 3's per-GPU training step (PVDS_PUNet, 8 patches x 2048 points = global batch 64 over
    8 GPUs, MSE bridge loss, grad clip 1.0, AdamW) on this one GPU, the reference's order of operations
    (train.py:107-143). ms_per_step = the captured step on a resident batch; ms_per_step_with_align = the step the reference's
    PU-Net loop actually runs (train.py:72-82: every batch is auction-aligned first, 100 rounds) -- fresh synthetic batches with
    the clean patch in random order, batch k + 1 aligned on a side stream as its own hipGraph while step k runs
    (train.AlignedBatches); ms_per_step_with_align_serial = the same with alignment and step one after the other.
    dense FLOPs of a step = 3 x forward (forward, data gradient, weight gradient) = 3 x 61.35 GFLOP x N / 8192 per patch
    (SURVEY 8d); the forward runs in fused.conv_math(), the data-gradient pass in bf16x6, the weight-gradient GEMMs in
    P2PB_TRAIN_MATH (bf16x3)."""
    import copy

    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.synthetic import synthetic_patches

    cfg = copy.deepcopy(PVDS)
    cfg["data"]["npoints"] = N
    torch.manual_seed(0)
    model = product.build_model(cfg, device="cuda")
    model.train()
    from p2p_bridge_amd import train as T

    tcfg = copy.deepcopy(cfg)
    tcfg["training"] = copy.deepcopy(T.PVDS_PUNET_TRAIN["training"])
    x1, x0 = synthetic_patches(B, N, seed=0)
    x1, x0 = x1.cuda(), x0.cuda()

    def timed(step_fn, n_warm, n):
        for _ in range(n_warm):
            loss = step_fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step_fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, float(loss)

    # eager: zero_grad -> loss -> backward -> clip + AdamW (optim.ClipAdamW, three launches) -> scheduler -> EMA
    opt, sched = T.load_optim_sched(tcfg, model, fused=True, skip_nonfinite=True)

    def eager():
        opt.zero_grad(set_to_none=True)
        loss = model(x0, x1)
        loss.backward()
        opt.step()
        sched.step()
        if model.ema is not None:
            model.ema.update()
        return loss.detach()

    eager_dt, _ = timed(eager, warmup, steps)
    # the same step as one captured hipGraph (train.GraphedStep, `python -m p2p_bridge_amd.train --graph`)
    stepper = T.GraphedStep(model, opt, sched, warmup=1)
    dt, loss = timed(lambda: stepper(x0, x1), warmup + 2, steps)
    # the same step fed by the PU-Net loop's data side: auction alignment of every batch (a23)
    tcfg["data"] = dict(tcfg.get("data", {}), dataset="PUNet")
    align = T.make_align_fn()
    # (a pool of device-resident batches, cycled: the host-side synthesis of a batch and its pageable host-to-device copy are a
    #  data loader's business -- worker processes, pinned memory -- and would otherwise be timed as part of the step)
    import itertools

    gen = T.synthetic_punet_batches(B, N, 100, "cuda")
    pool = [next(gen) for _ in range(8)]
    torch.cuda.synchronize()
    feed = T.AlignedBatches(itertools.cycle(pool), tcfg, align, capture=True)

    def with_align():
        d = next(feed)
        return stepper(d["x_gt"], d["x_start"], d["x_cond"])

    dt_al, _ = timed(with_align, warmup + 2, steps)
    raw = itertools.cycle(pool)

    def with_align_serial():
        d = T.get_data_batch(next(raw), tcfg, align)
        return stepper(d["x_gt"], d["x_start"], d["x_cond"])

    dt_ser, _ = timed(with_align_serial, 2, steps)
    flop = 3.0 * 61.35e9 * N / 8192.0 * B
    return {"workload": f"PVDS_PUNet training step, {B} patches x {N} points per GPU (BASELINE configs[2] = global batch 64 on 8 GPUs), "
                        "mse bridge loss, clip 1.0, AdamW, scheduler, EMA; hand-written forward / backward / optimiser kernels, the step "
                        "captured as one hipGraph (train.GraphedStep); eager_ms_per_step = the same step launched eagerly",
            "ms_per_step": round(dt * 1e3, 2), "ms_per_step_with_align": round(dt_al * 1e3, 2),
            "ms_per_step_with_align_serial": round(dt_ser * 1e3, 2), "eager_ms_per_step": round(eager_dt * 1e3, 2), "patches_per_s": round(B / dt, 1), "points_per_s": round(B * N / dt, 1),
            "dense_tflops": round(flop / dt / 1e12, 2), "frac_of_f16x3_ceiling": round(flop / dt / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 3), 4),
            "frac_of_bf16x6_ceiling": round(flop / dt / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 6), 4), "steps": steps, "warmup": warmup,
            "final_loss": round(loss, 5)}


def alt_math_leg(cfg, sd, x_start, args):
    """NOT `value`: the same workload in the other split arithmetic (bf16x6: six products, no range contract), timed the
    same way on a fresh model (its own captured graph), and how far ONE network evaluation on identical inputs is from the
    default's (a free-running 30-step sampler amplifies any difference through neighbour-index decisions, so its end
    points are not comparable)"""
    from p2p_bridge_amd import fused, p2pb as product

    fused.set_conv_math("bf16x6")
    try:
        model = product.build_model(cfg, sd, device=str(x_start.device))
        run = lambda: model.sample(x_start=x_start, steps=args.T, log_count=1, verbose=False, graph=bool(args.graph))
        for _ in range(max(1, args.warmup)):
            out = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.full((x_start.shape[0],), 500.0, device=x_start.device)
        model.eval()
        with torch.no_grad():
            e6 = model.model(x_start, t).clone()
            fused.set_conv_math(None)
            e3 = model.model(x_start, t)
        diff = float((e3 - e6).abs().max().item())
        assert torch.isfinite(out["x_pred"]).all()
    finally:
        fused.set_conv_math(None)
    return {"conv_math": "bf16x6 (P2PB_CONV_MATH=bf16x6; not the headline)",
            "value": round(args.batch * args.points * args.steps / dt, 1), "unit": "points/s",
            "ms_per_step": round(dt / args.steps * 1e3, 2), "max_abs_diff_of_one_evaluation_vs_default": diff}


def pvdl_leg(B=8, N=50000, T=30, extra=3, reps=2, fit_batch=64):
    """NOT `value`: BASELINE config 4 on this one GPU -- full-width PVDL (118.6 M parameters, channels 64..1024, 13 PVConvs),
    xyz + RGB condition, B clouds of 50000 points, T = 30, the product's own conditional sampler as one hipGraph per step.
    dense-equivalent FLOPs: SURVEY 8d, 486 GFLOP per sample and evaluation at 50000 points. At this batch an evaluation is bound
    by the level-0 farthest-point sampling (12499 dependent rounds per cloud, one workgroup each); batch hides it
    (profiles/r03d_pvdl_large_batches.txt)."""
    import copy

    from p2p_bridge_amd import p2pb as product
    from p2p_bridge_amd.synthetic import synthetic_patches

    c = copy.deepcopy(PVDS)
    c["data"]["npoints"] = N
    c["diffusion"]["beta_end"] = 3e-4
    c["model"]["extra_feature_channels"] = extra
    c["model"]["dropout"] = 0.1
    c["model"]["PVD"].update(feat_embed_dim=64, attention_heads=12, channels=[64, 128, 256, 512, 1024],
                             n_sa_blocks=[2, 3, 2, 2], n_fp_blocks=[2, 3, 2, 2])
    torch.manual_seed(0)
    model = product.build_model(c, device="cuda")

    def timed(Bx, nrep):
        x, _ = synthetic_patches(Bx, N, seed=1)
        g = torch.Generator().manual_seed(2)
        cond = torch.rand(Bx, extra, N, generator=g)
        x, cond = x.cuda(), cond.cuda()
        run = lambda: model.sample(x_start=x, x_cond=cond, steps=T, log_count=1, verbose=False, graph=True)
        out = run()  # weight packs + capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nrep):
            out = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nrep
        assert torch.isfinite(out["x_pred"]).all()
        return dt

    dt = timed(B, reps)
    tf = 486.0 * B * T / dt / 1e3
    res = {"workload": f"PVDL_SNPP xyz+RGB ({3 + extra}-ch), {B} x {N}-pt clouds, T={T}, hipGraph (BASELINE configs[3] on one GPU)",
           "value": round(B * N / dt, 1), "unit": "points/s", "ms_per_sample_call": round(dt * 1e3, 1),
           "ms_per_evaluation": round(dt * 1e3 / T, 2), "dense_equivalent_tflops": round(tf, 1),
           "frac_of_split_ceiling": round(tf / split_peak_tflops(), 4), "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if fit_batch:
        # BASELINE configs[3] says "as many as fit": the B = 8 figure above is the LATENCY of the path (an evaluation is the level-0
        # farthest-point sampling's 12499 dependent rounds, one workgroup per cloud); the batch this part's memory takes hides it
        # (profiles/r03d_pvdl_large_batches.txt). One timed call at `fit_batch` clouds (two sampler chains, the default there).
        model.clear_graphs()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        dtf = timed(fit_batch, 1)
        tff = 486.0 * fit_batch * T / dtf / 1e3
        res["value_at_fit"] = {"clouds": fit_batch, "value": round(fit_batch * N / dtf, 1), "unit": "points/s",
                               "ms_per_evaluation": round(dtf * 1e3 / T, 2), "dense_equivalent_tflops": round(tff, 1),
                               "frac_of_split_ceiling": round(tff / split_peak_tflops(), 4),
                               "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    return res


def result_line(args, world, dt, dist):
    pts = world * args.batch * args.points * args.steps
    return {
        "metric": "denoised points/sec (8192-pt patches, T=30)", "value": round(pts / dt, 1), "unit": "points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PVDS_PUNet xyz-only, {args.points}-pt patches, T={args.T} bridge steps, batch "
                               f"{args.batch} per GPU (BASELINE configs[1])", "patches_per_gpu": args.batch,
                   "points_per_patch": args.points, "bridge_steps": args.T, "parallelism": f"patch-shard x{world}",
                   "process_group_world_size": dist.get_world_size() if dist is not None else 1,
                   "process_group_backend": (dist.get_backend() if dist is not None else None),
                   "hipgraph": bool(args.graph), "weights": "seeded random init (26.44 M params)"},
    }


def dry_run(args, dist, rank, world):
    """the N-rank protocol without a GPU (tests/test_bench_launch.py): same barrier / max-over-ranks bracket around a
    stand-in step whose duration depends on the rank, same JSON line; `value` is meaningless and flagged as such"""
    from p2p_bridge_amd import sharding

    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (rank + 1))
    dt_own = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt_local)
    res = result_line(args, world, dt, dist)
    res["ranks"] = sharding.rank_evidence(dt_own, args.batch * args.points * args.steps, None)
    res["dry_run"] = True
    res["data"] = "none (dry run: launch / rendezvous / timing protocol only)"
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
