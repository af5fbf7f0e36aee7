"""tarok_amd — MI355X-native vectorised Tarok card-play environment.

Accelerates one hot path of anzeA/Tarok: lock-stepping N
independent 4-player games — legal-card masks, trick resolution and
Klop / Berac / Navadna_igra scoring — as hand-written HIP kernels behind the
C ABI of include/tarok_env.h.  Importing the package is cheap and works
without a GPU; creating an environment does not (no CPU fallback)."""
from . import karte
from ._native import TarokNativeError, build

__all__ = ["karte", "build", "TarokNativeError", "TarokVecEnv", "Obs"]


def __getattr__(name):
    if name in ("TarokVecEnv", "Obs"):
        from . import env
        return getattr(env, name)
    raise AttributeError(name)
