"""ONE parser for the A/B and experiment switches of the package: `P2PB_EXPERIMENT="key=value;key=value"` (read at every query, so a
test can change it at run time; the library's C side parses the same variable once per site: csrc/abi.hip p2pb_experiment_long).

The switches a USER may need keep their own variables: P2PB_LIB_PATH, P2PB_CONV_MATH, P2PB_TRAIN_MATH,
P2PB_SAMPLE_CHAINS, P2PB_SEGMENTED_BACKWARD, P2PB_FUSED_OPTIM, P2PB_F16_OVERFLOW (and bench.py's P2PB_CPU_THREADS / P2PB_CPU_PROCS).

Keys (default): conv_pre (8,16,32:8,16), compact (16:16), wide_f16_min_cin (16), prepass_blocks (9), sa_gather (1), fps_big (grid),
nn_cells (1), wgrad_overlap (0), dgrad_math (follows P2PB_TRAIN_MATH), sparse_wgrad_min_r (16), chain_stagger_pct (0 | 100 by cloud size) -- Python side; conv_wide_min (256), am_chunks (auto), pw_wm (auto), pw_pp (1), fps_mid (512),
fps_coop_test_fallback (0), vox_onepass (per shape), devox_cl4 (1) -- library side."""
import os


def _table():
    out = {}
    for item in os.environ.get("P2PB_EXPERIMENT", "").split(";"):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip().lower()] = v.strip()
    return out


def get(key, default=None):
    return _table().get(key, default)


def get_int(key, default):
    v = get(key)
    return int(v) if v not in (None, "") else default


def setting(**kv):
    """the P2PB_EXPERIMENT string with `kv` merged in (None removes a key): for tests / tools that build an environment"""
    t = _table()
    for k, v in kv.items():
        if v is None:
            t.pop(k, None)
        else:
            t[k] = str(v)
    return ";".join(f"{k}={v}" for k, v in t.items())
