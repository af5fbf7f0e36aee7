"""Build + ctypes binding of libtarokenv.so (include/tarok_env.h).

The HIP library is the product: there is NO CPU fallback.  Anything that needs
the environment raises if the shared library is missing or no GPU is visible."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "tarok_env.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "tarok_device.h"), os.path.join(HERE, "csrc", "deal_network.inc"),
        os.path.join(HERE, "csrc", "tarok_learner.inc"),
        os.path.join(ROOT, "include", "tarok_env.h")]
LIB_PATH = os.environ.get("TAROK_LIB") or os.path.join(HERE, "libtarokenv.so")   # TAROK_LIB: A/B diagnostics only
ARCH = "gfx950"

SYMBOLS = [
    "tarok_strerror", "tarok_abi_version", "tarok_device_count", "tarok_last_hip_error",
    "tarok_create", "tarok_destroy", "tarok_num_games", "tarok_set_option", "tarok_reset", "tarok_exchange",
    "tarok_legal_actions", "tarok_step", "tarok_prefetch", "tarok_policy_random", "tarok_step_random",
    "tarok_run_random", "tarok_krog_random", "tarok_rollout_random", "tarok_get_state", "tarok_set_state", "tarok_get_counters", "tarok_debug_stamps", "tarok_debug_stamps_sized", "tarok_debug_refill_selftest", "tarok_observe", "tarok_sample_policy", "tarok_policy_mlp", "tarok_policy_step", "tarok_expand_features", "tarok_ppo_loss",
    "tarok_targets_ref", "tarok_learn_returns", "tarok_learn_chain", "tarok_learn_workspace_bytes", "tarok_learn_dw", "tarok_learn_adam",
    "tarok_observe_ref", "tarok_observe_exchange_ref", "tarok_observe_hands_ref", "tarok_get_history", "tarok_set_history",
]


class TarokNativeError(RuntimeError):
    pass


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    return "hipcc"


def needs_build():
    if os.environ.get("TAROK_LIB"):          # an A/B library is whatever its builder made it: never rebuilt over
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("TAROK_LIB=%s does not exist" % LIB_PATH)
        return False
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> tarok_amd/libtarokenv.so (in-tree).

    Safe when several ranks of one node start together: one process compiles (exclusive lock on a
    side file), into a temporary name that is renamed over the library in one step; the others
    wait, find the library current and return."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
                # -amdgpu-kernarg-preload-count: the first kernel arguments arrive in scalar registers with the dispatch
                # instead of behind a scalar load at the top of the kernel (0.1-0.2 us of every latency-bound launch:
                # profiles/r04_ab_step.txt); firmware without the feature runs the kernels' own load prologue
                cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC",
                       "-mllvm", "-amdgpu-kernarg-preload-count=16",
                       "-I", os.path.join(ROOT, "include"), "-o", tmp, SRC]
                if verbose:
                    print(" ".join(cmd))
                try:
                    subprocess.check_call(cmd)
                    # gfx950 erratum gate (isa_check.py): a 64-bit shift whose amount sits in the last VGPR its kernel
                    # allocates shifts by v0 now and then — such a library is never installed
                    from . import isa_check
                    bad, _ = isa_check.check(tmp)
                    if bad:
                        raise TarokNativeError("the compiled library has 64-bit shifts with the amount in the last allocated "
                                               "VGPR (gfx950 erratum, DESIGN.md section 3):\n" + isa_check.describe(bad))
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _check_single_hip_runtime():
    """torch bundles its own libamdhip64 (same soname as /opt/rocm's).  The
    loader shares it with us when torch is imported first; two runtimes in one
    process would make torch's pointers/streams meaningless to our launches."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(os.path.realpath(line.split()[-1]))
    except OSError:
        return
    if len(paths) > 1:
        raise TarokNativeError("two HIP runtimes are mapped (%s): import torch before tarok_amd" % sorted(paths))


_lib = None


def lib():
    """Load libtarokenv.so (importing torch first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must precede the dlopen below)
    if not os.path.exists(LIB_PATH):
        raise TarokNativeError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    _check_single_hip_runtime()
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32
    L.tarok_strerror.restype = C.c_char_p; L.tarok_strerror.argtypes = [i32]
    L.tarok_abi_version.restype = i32; L.tarok_abi_version.argtypes = []
    L.tarok_device_count.restype = i32; L.tarok_device_count.argtypes = []
    L.tarok_last_hip_error.restype = i32; L.tarok_last_hip_error.argtypes = []
    L.tarok_create.restype = i32; L.tarok_create.argtypes = [C.POINTER(vp), i32, i64, u64, u64, i32, i32]
    L.tarok_destroy.restype = None; L.tarok_destroy.argtypes = [vp]
    L.tarok_num_games.restype = i64; L.tarok_num_games.argtypes = [vp]
    if hasattr(L, "tarok_set_option"):      # (absent from older libraries loaded through TAROK_LIB for A/B runs)
        L.tarok_set_option.restype = i32; L.tarok_set_option.argtypes = [vp, i32, i32]
    L.tarok_reset.restype = i32; L.tarok_reset.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, i32, vp]
    L.tarok_exchange.restype = i32; L.tarok_exchange.argtypes = [vp, vp, vp, vp]
    L.tarok_legal_actions.restype = i32; L.tarok_legal_actions.argtypes = [vp, vp, vp, vp]
    L.tarok_step.restype = i32; L.tarok_step.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
    L.tarok_prefetch.restype = i32; L.tarok_prefetch.argtypes = [vp, vp]
    L.tarok_policy_random.restype = i32; L.tarok_policy_random.argtypes = [vp, vp, vp, vp]
    L.tarok_step_random.restype = i32; L.tarok_step_random.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
    L.tarok_run_random.restype = i32; L.tarok_run_random.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, i32, vp]
    L.tarok_krog_random.restype = i32; L.tarok_krog_random.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, i32, vp]
    L.tarok_rollout_random.restype = i32; L.tarok_rollout_random.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.tarok_get_state.restype = i32; L.tarok_get_state.argtypes = [vp, vp, vp]
    L.tarok_observe.restype = i32; L.tarok_observe.argtypes = [vp, vp, vp]
    L.tarok_sample_policy.restype = i32; L.tarok_sample_policy.argtypes = [vp, vp, vp, vp, vp, vp]
    L.tarok_policy_mlp.restype = i32; L.tarok_policy_mlp.argtypes = [vp] * 14
    L.tarok_policy_step.restype = i32; L.tarok_policy_step.argtypes = [vp] * 16 + [i32, vp]
    L.tarok_expand_features.restype = i32; L.tarok_expand_features.argtypes = [vp, i64, vp, vp, vp, vp]
    f32 = C.c_float
    L.tarok_ppo_loss.restype = i32; L.tarok_ppo_loss.argtypes = [vp, i64] + [vp] * 7 + [f32] * 3 + [vp] * 4
    if hasattr(L, "tarok_targets_ref"):
        L.tarok_targets_ref.restype = i32; L.tarok_targets_ref.argtypes = [vp, i32] + [vp] * 6 + [f32] + [vp] * 3
    if hasattr(L, "tarok_learn_chain"):
        L.tarok_learn_returns.restype = i32; L.tarok_learn_returns.argtypes = [vp, i32] + [vp] * 6 + [f32] + [vp] * 4
        L.tarok_learn_chain.restype = i32; L.tarok_learn_chain.argtypes = [vp, i64] + [vp] * 4 + [f32] * 3 + [vp] * 18
        L.tarok_learn_workspace_bytes.restype = i64; L.tarok_learn_workspace_bytes.argtypes = [vp]
        L.tarok_learn_dw.restype = i32; L.tarok_learn_dw.argtypes = [vp, i64] + [vp] * 10
        L.tarok_learn_adam.restype = i32; L.tarok_learn_adam.argtypes = [vp] * 6 + [f32] * 5 + [vp] * 6 + [i32, vp]
    L.tarok_observe_ref.restype = i32; L.tarok_observe_ref.argtypes = [vp, vp, vp, vp]
    L.tarok_observe_exchange_ref.restype = i32; L.tarok_observe_exchange_ref.argtypes = [vp, vp, vp]
    L.tarok_observe_hands_ref.restype = i32; L.tarok_observe_hands_ref.argtypes = [vp, vp, vp]
    L.tarok_get_history.restype = i32; L.tarok_get_history.argtypes = [vp, vp, vp]
    L.tarok_set_history.restype = i32; L.tarok_set_history.argtypes = [vp, vp, vp]
    L.tarok_debug_stamps.restype = i32; L.tarok_debug_stamps.argtypes = [vp, vp]
    if hasattr(L, "tarok_debug_stamps_sized"):      # (absent from older libraries loaded through TAROK_LIB for A/B runs)
        L.tarok_debug_stamps_sized.restype = i32; L.tarok_debug_stamps_sized.argtypes = [vp, vp, i64]
    if hasattr(L, "tarok_debug_refill_selftest"):
        L.tarok_debug_refill_selftest.restype = i32; L.tarok_debug_refill_selftest.argtypes = [vp, i32, i32, u32, i32, i32, vp]
    L.tarok_set_state.restype = i32; L.tarok_set_state.argtypes = [vp, vp, vp]
    L.tarok_get_counters.restype = i32; L.tarok_get_counters.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def check(code):
    if code != 0:
        L = lib()
        raise TarokNativeError("libtarokenv: %s (code %d, hipError %d)" % (
            L.tarok_strerror(code).decode(), code, L.tarok_last_hip_error()))
