"""ONE parser for the A/B and experiment switches of the package: `P2PB_EXPERIMENT="key=value;key=value"` (read at every query, so a
test can change it at run time; the library's C side parses the same variable once per site: csrc/abi.hip p2pb_experiment_long).

The switches a USER may need keep their own variables: P2PB_LIB_PATH, P2PB_CONV_MATH, P2PB_TRAIN_MATH,
P2PB_SAMPLE_CHAINS, P2PB_SEGMENTED_BACKWARD, P2PB_FUSED_OPTIM, P2PB_F16_OVERFLOW (and bench.py's P2PB_CPU_THREADS / P2PB_CPU_PROCS).

Keys (default): conv_pre (8,16,32:8,16), compact (16:16), wide_f16_min_cin (16), prepass_blocks (9), sa_gather (1), fps_big (grid),
nn_cells (1), wgrad_overlap (0), dgrad_math (follows P2PB_TRAIN_MATH), sparse_wgrad_min_r (16), chain_stagger_pct (0 | 100 by cloud size), row_max (1), train_fold (1) -- Python side; conv_wide_min (256), am_chunks (auto), pw_wm (auto), pw_pp (1), fps_mid (512),
fps_coop_test_fallback (0), vox_onepass (per shape), devox_cl4 (1) -- library side."""
import os
import warnings

# Switches that had their own environment variable before round 5 and are keys of P2PB_EXPERIMENT now. Setting one of the old
# names used to change behaviour and would now be ignored SILENTLY: warn once per process, naming the key to use.
LEGACY = {"P2PB_CONV_PRE": "conv_pre", "P2PB_COMPACT": "compact", "P2PB_FPS_BIG": "fps_big", "P2PB_FPS_COOP": "fps_big",
          "P2PB_NN_CELLS": "nn_cells", "P2PB_SA_GATHER": "sa_gather", "P2PB_WGRAD_OVERLAP": "wgrad_overlap",
          "P2PB_VOX_ONEPASS": "vox_onepass", "P2PB_PW_PP": "pw_pp", "P2PB_PW_WM": "pw_wm", "P2PB_AM_CHUNKS": "am_chunks",
          "P2PB_PREPASS_BLOCKS": "prepass_blocks", "P2PB_DEVOX_CL4": "devox_cl4", "P2PB_FPS_MID": "fps_mid"}
_warned = False


def warn_legacy():
    """called when the package is imported and by every query until it has fired once"""
    global _warned
    if _warned:
        return
    stale = [k for k in LEGACY if k in os.environ]
    if stale:
        _warned = True
        warnings.warn("ignored environment switch(es) " + ", ".join(f"{k} (now P2PB_EXPERIMENT=\"{LEGACY[k]}=...\")" for k in stale)
                      + ": they became keys of P2PB_EXPERIMENT in round 5", RuntimeWarning, stacklevel=3)


def _table():
    warn_legacy()
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
