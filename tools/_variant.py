"""Helper of the A/B tools: load another build of libbcplan (possibly an older one that lacks the newest entry points)."""
import ctypes as C
import os


def use_lib(path):
    import torch  # noqa: F401  (first: its HIP runtime must be the one in the process, whatever the variant links against)
    from bc_gym_planning_env_amd import _lib
    if path in ('', '-', None):
        return _lib.LIB_PATH
    _lib.LIB_PATH = os.path.abspath(path)
    probe = C.CDLL(_lib.LIB_PATH)
    for name in list(_lib.SYMBOLS):
        if not hasattr(probe, name):
            del _lib.SYMBOLS[name]     # (tools only: the product's loader insists on every symbol)
    return _lib.LIB_PATH
