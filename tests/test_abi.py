"""The C-ABI shared library loads without a GPU and exports every symbol include/p2pb_hip.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "p2pb_hip.h")


def header_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p2pb_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from p2p_bridge_amd import _lib, build

    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/p2pb_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms
    assert _lib.lib().p2pb_target_arch() == b"gfx950"


def test_library_contains_gfx950_code_object():
    from p2p_bridge_amd import _lib

    out = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_kernel_issues_ds_write_b96():
    """Round 4's cross-stream corruption sat in a kernel whose LDS staging the compiler had merged into
    ds_write_b96 + ds_write2_b32; the mechanism is unexplained (barrier and s_waitcnt were in order in the ISA), so the invariant
    that protects the library is enforced here: the disassembly of every translation unit of libp2pb_hip.so holds no 12-byte LDS
    store. (tests/test_concurrency_gpu.py is the dynamic half of the net.)"""
    import glob
    import tempfile

    from p2p_bridge_amd import build

    build.build()
    llvm = "/opt/rocm/lib/llvm/bin"
    objs = sorted(glob.glob(os.path.join(build.OBJ, "*.o")))
    assert len(objs) >= 16
    seen_b128 = 0
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            fat, co = os.path.join(tmp, "x.fat"), os.path.join(tmp, "x.co")
            subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", o], check=True)
            subprocess.run([f"{llvm}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            isa = subprocess.run([f"{llvm}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
            assert "s_endpgm" in isa, o
            seen_b128 += isa.count("ds_write_b128")
            for bad in ("ds_write_b96", "ds_store_b96"):
                assert bad not in isa, (bad, os.path.basename(o))
    assert seen_b128 > 100  # (the grep sees LDS stores at all)


def test_no_cpu_fallback():
    """product ops refuse CPU tensors (the oracle is never a fallback)."""
    import torch

    from p2p_bridge_amd import pointnet2_batch_cuda as ext

    with pytest.raises(RuntimeError):
        ext.furthest_point_sampling_forward(torch.zeros(1, 3, 16), 4)
    with pytest.raises(RuntimeError):
        ext.avg_voxelize_forward(torch.zeros(1, 2, 16), torch.zeros(1, 3, 16, dtype=torch.int32), 4)


def test_product_never_imports_oracle():
    import glob

    for f in glob.glob(os.path.join(ROOT, "p2p_bridge_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
        assert "cpu_ops" not in src, f


def test_split_terms_thread_override_is_per_thread():
    """p2pb_set_split_terms_thread (fused.split_math): a temporary arithmetic switch on one thread is invisible to the
    others; the process-wide setter still reaches every thread without an override (no GPU needed: host state only)"""
    import threading

    from p2p_bridge_amd import _lib

    lib = _lib.lib()
    base = lib.p2pb_get_split_terms()
    seen = {}

    def worker():
        assert lib.p2pb_set_split_terms_thread(6) == 0
        seen["inside"] = lib.p2pb_get_split_terms()
        ev1.set()
        ev2.wait(5)
        assert lib.p2pb_set_split_terms_thread(0) == 0
        seen["cleared"] = lib.p2pb_get_split_terms()

    ev1, ev2 = threading.Event(), threading.Event()
    t = threading.Thread(target=worker)
    t.start()
    ev1.wait(5)
    seen["other_thread"] = lib.p2pb_get_split_terms()
    ev2.set()
    t.join()
    assert seen == {"inside": 6, "other_thread": base, "cleared": base}
    assert lib.p2pb_set_split_terms_thread(5) != 0  # not an arithmetic


def test_abi_version_matches_header_and_binding():
    """p2pb_version() used to return 1 through every ABI change. Header macro == library == binding, and the loader
    refuses a library of another version (a stale .so behind P2PB_LIB_PATH) with a message instead of failing late."""
    from p2p_bridge_amd import _lib, build

    build.build()
    macro = int(re.search(r"#define\s+P2PB_ABI_VERSION\s+(\d+)", open(HDR).read()).group(1))
    assert macro == _lib.ABI_VERSION
    assert ctypes.CDLL(_lib.LIB_PATH).p2pb_version() == macro
    code = ("import p2p_bridge_amd._lib as L\n"
            "L.ABI_VERSION += 1\n"
            "try:\n    L.lib()\nexcept L.P2PBError as e:\n    assert 'ABI version' in str(e), e\n    print('refused')\n")
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "refused" in r.stdout, r.stderr[-2000:]


def test_experiment_switch_parsers_agree_and_legacy_names_warn():
    """the C parser (csrc/abi.hip p2pb_experiment_long) reads P2PB_EXPERIMENT like the Python one
    (p2p_bridge_amd/_experiment.py; the C function is internal to the library, reached by its mangled name): case-insensitive keys, blanks around key, '=' and value; and a pre-round-5 switch name
    in the environment (silently ignored since) raises a RuntimeWarning naming its replacement."""
    code = r'''
import ctypes, os, warnings
os.environ["P2PB_EXPERIMENT"] = " Pw_Pp = 0 ;conv_wide_min=77; FPS_MID =9;x=1"
from p2p_bridge_amd import _lib, _experiment
so = ctypes.CDLL(_lib.LIB_PATH)
f = getattr(so, "_Z20p2pb_experiment_longPKcl"); f.restype = ctypes.c_long; f.argtypes = [ctypes.c_char_p, ctypes.c_long]
for key, dflt in (("pw_pp", 1), ("conv_wide_min", 256), ("fps_mid", 512), ("absent", 5), ("pw", 3), ("x", 0)):
    assert f(key.encode(), dflt) == _experiment.get_int(key, dflt), (key, f(key.encode(), dflt), _experiment.get_int(key, dflt))
assert f(b"pw_pp", 1) == 0 and f(b"fps_mid", 512) == 9 and f(b"pw", 3) == 3
os.environ["P2PB_FPS_BIG"] = "coop"
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    _experiment.get("fps_big")
    _experiment.get("fps_big")
assert len(w) == 1 and "fps_big" in str(w[0].message) and issubclass(w[0].category, RuntimeWarning), [str(x.message) for x in w]
print("ok")
'''
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-2000:]
