"""Timing of steps with the diagnostic flags of a -DBCP_DIAG build (ablation switches: results wrong by construction).
The shipping Python layer knows nothing about these flags; the tools call the library's timing loop themselves."""
import ctypes as C

from bc_gym_planning_env_amd import _lib


def time_steps_with_flags(env, actions, steps, extra_flags=0, noise_z=None):
    """Average device time (ms) of one step over `steps` back-to-back steps with `extra_flags` OR-ed into the step flags."""
    a, io, flags = env._timing_io(actions, noise_z)
    ms = C.c_float()
    _lib.check(env._lib.bcp_time_steps(env._h, C.byref(io), flags | int(extra_flags), int(steps), env._stream(), C.byref(ms)))
    return ms.value
