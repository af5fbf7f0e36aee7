"""CPU-side checks of the boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/tarok_env.h declares; the package imports without a GPU
and refuses to run without one (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    import torch  # noqa: F401  -- before any dlopen of libtarokenv.so: both must share ONE HIP runtime
    import tarok_amd
    return tarok_amd.build()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "tarok_env.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tarok_[a-z_]+)\s*\(", src)))


def test_header_symbols_all_exported(libpath):
    from tarok_amd import _native
    L = ctypes.CDLL(libpath)
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libtarokenv.so does not export %s" % n
    assert set(names) == set(_native.SYMBOLS)
    L.tarok_abi_version.restype = ctypes.c_int
    assert L.tarok_abi_version() == 5
    L.tarok_strerror.restype = ctypes.c_char_p
    assert L.tarok_strerror(0) == b"ok" and b"argument" in L.tarok_strerror(-1)


def test_code_object_is_gfx950(libpath):
    data = open(libpath, "rb").read()
    assert b"gfx950" in data
    assert b"gfx942" not in data and b"sm_" not in data


def test_no_gpu_means_loud_failure(libpath):
    import torch
    import tarok_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(tarok_amd.TarokNativeError):
        tarok_amd.TarokVecEnv(16)


def test_create_rejects_bad_arguments(libpath):
    from tarok_amd import _native
    L = _native.lib()
    h = ctypes.c_void_p()
    assert L.tarok_create(ctypes.byref(h), 0, 0, 0, 0, 0, 0) == -1          # n_games = 0
    assert L.tarok_create(ctypes.byref(h), 0, 16, 0, 0, 7, 0) == -1         # unknown mix
    if L.tarok_device_count() == 0:
        assert L.tarok_create(ctypes.byref(h), 0, 16, 0, 0, 0, 0) == -4     # TAROK_ENODEV
    assert L.tarok_num_games(None) == 0


def test_round3_entry_points_reject_bad_arguments_without_a_gpu(libpath):
    """The entry points added in round 3 validate their arguments before any HIP call (no compute without a GPU)."""
    from tarok_amd import _native
    L = _native.lib()
    assert L.tarok_set_option(None, 2, 4) == -1
    assert L.tarok_learn_workspace_bytes(None) == 0
    z = ctypes.c_void_p(0)
    assert L.tarok_targets_ref(None, 48, z, z, z, z, z, z, 0.1, z, z, z) == -1
    assert L.tarok_learn_returns(None, 48, z, z, z, z, z, z, 1.0, z, z, z, z) == -1
    assert L.tarok_learn_dw(None, 128, z, z, z, z, z, z, z, z, z, z) == -1
    assert L.tarok_learn_dw(None, 1 << 23, z, z, z, z, z, z, z, z, z, z) == -1         # (also past TAROK_LEARN_MAX_BATCH)
    assert L.tarok_learn_adam(None, z, z, z, z, z, 1e-3, 0.9, 0.999, 1e-8, 1.0, z, z, z, z, z, z, 1, z) == -1


def test_product_never_touches_the_oracle():
    """tarok_amd/ and bench.py's GPU legs must not import or link oracle/."""
    pkg = os.path.join(ROOT, "tarok_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libtarok_oracle" not in text, f
                assert "tarok_oracle.h" not in text, f
