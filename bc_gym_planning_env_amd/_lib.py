"""ctypes binding of libbcplan.so (include/bcplan.h).  There is no fallback: if the HIP library is missing or no
GPU is visible, creating an env raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbcplan.so")

ABI_VERSION = 2
MAX_VERTS = 32
MODEL_TRICYCLE, MODEL_DIFFDRIVE = 0, 1
REWARD_CONTINUOUS, REWARD_PURE_PURSUIT = 0, 1
STEP_AUTO_RESET, STEP_ACTIONS_F32 = 1, 2
ERR_ANGLE_JUMP, ERR_TIME_ORDER, ERR_INTERNAL = 1, 2, 4
TUNE_EXACT_MODE, TUNE_DENSE_THRESHOLD, TUNE_CULL, TUNE_DEFER, TUNE_EDT_LDS, TUNE_FUSED, TUNE_EGO_SPARSE, TUNE_NEAR_DILATE = 0, 1, 2, 3, 4, 5, 6, 7
TUNE_LOCAL_PAIRS, TUNE_EGO_LIST_STRIDE, TUNE_NEAR_SHIFT = 8, 9, 10
E_NO_DEVICE = -2
OPT_DIFFDRIVE_NOISE = 1
EGO_KERNELS = {0: "none", 1: "ego_sparse_kernel", 2: "ego_costmap_kernel<staged>", 3: "ego_costmap_binned_kernel",
               4: "ego_costmap_window_kernel", 5: "ego_costmap_kernel<global>"}

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


class BcpParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("n_verts", C.c_int32), ("dynamic_model", C.c_int32),
        ("model_front_column_pid", C.c_int32), ("noise_on", C.c_int32), ("iteration_timeout", C.c_int32),
        ("options", C.c_int32),
        ("verts", (C.c_double * 2) * MAX_VERTS),
        ("dt", C.c_double), ("front_wheel_from_axis", C.c_double), ("max_front_wheel_angle", C.c_double),
        ("max_front_wheel_speed", C.c_double), ("max_linear_acceleration", C.c_double),
        ("max_angular_acceleration", C.c_double), ("front_column_p_gain", C.c_double),
        ("alpha", C.c_double * 6),
        ("spatial_precision", C.c_double), ("angular_precision", C.c_double),
        ("spatial_progress_multiplier", C.c_double),
        ("reward_provider", C.c_int32), ("control_delay", C.c_int32), ("pose_delay", C.c_int32),
        ("state_delay", C.c_int32),
    ]


class BcpState(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("angle", C.c_void_p), ("v", C.c_void_p), ("w", C.c_void_p),
        ("steering_motor_command", C.c_void_p), ("wheel_angle", C.c_void_p), ("min_spat_dist_so_far", C.c_void_p),
        ("target_idx", C.c_void_p), ("current_iter", C.c_void_p), ("robot_collided", C.c_void_p),
        ("pose_seen", C.c_void_p), ("robot_state_seen", C.c_void_p), ("control_queue", C.c_void_p),
        ("poses_queue", C.c_void_p), ("robot_state_queue", C.c_void_p),
    ]


class BcpMiniWorldParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "inner_h", "inner_w", "mid_margin", "out_margin", "min_obstacle_angle", "max_obstacle_angle", "lim_euc_dist",
        "lim_ang_dist", "angular_pose_noise_scale", "resolution", "goal_spat_dist", "goal_ang_dist")]


class BcpStepIO(C.Structure):
    _fields_ = [
        ("actions", C.c_void_p), ("noise_z", C.c_void_p), ("noise_z_out", C.c_void_p), ("reward", C.c_void_p),
        ("done", C.c_void_p), ("collided_now", C.c_void_p), ("err", C.c_void_p),
    ]


# every symbol include/bcplan.h declares: (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "bcp_last_error": (C.c_char_p, []),
    "bcp_abi_version": (C.c_int, []),
    "bcp_create": (C.c_int, [C.POINTER(BcpParams), C.c_int64, C.c_int, C.c_int64, C.POINTER(_H)]),
    "bcp_destroy": (C.c_int, [_H]),
    "bcp_seed": (C.c_int, [_H, C.c_uint64]),
    "bcp_set_tuning": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bcp_set_geometry_pool": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p]),
    "bcp_set_costmaps": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "bcp_set_paths": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "bcp_bind_state": (C.c_int, [_H, C.POINTER(BcpState)]),
    "bcp_bind_initial_state": (C.c_int, [_H, C.POINTER(BcpState)]),
    "bcp_reset_masked": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "bcp_broadcast_state": (C.c_int, [_H, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_step": (C.c_int, [_H, C.POINTER(BcpStepIO), C.c_uint32, C.c_void_p]),
    "bcp_rollout": (C.c_int, [_H, C.POINTER(BcpStepIO), C.c_int32, C.c_uint32, C.c_void_p]),
    "bcp_expired_waits": (C.c_int, [_H, C.POINTER(C.c_int64), C.c_void_p]),
    "bcp_parked_poses": (C.c_int, [_H, C.POINTER(C.c_int64), C.c_void_p]),
    "bcp_side_stream": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_void_p)]),
    "bcp_robot_step": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcp_pose_collides": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_is_robot_colliding": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_is_footprint_colliding": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_uint8,
                                             C.c_void_p, C.c_void_p]),
    "bcp_reward": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "bcp_find_last_reached": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_path_velocity": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcp_pixel_footprint": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    "bcp_pack_mask_bits": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_unpack_mask_bits": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_normalize_angle": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bcp_world_to_pixel": (C.c_int, [_H, C.c_void_p, C.c_int64, _f64p, C.c_double, C.c_void_p, C.c_void_p]),
    "bcp_egocentric_shape": (C.c_int, [_H, _f64p, _i32p]),
    "bcp_egocentric_costmaps": (C.c_int, [_H, C.c_void_p, C.c_int64, _f64p, _f64p, C.c_uint8, C.c_void_p,
                                          C.c_void_p]),
    "bcp_egocentric_route": (C.c_int, [_H, _i32p]),
    "bcp_goal_n_state": (C.c_int, [_H, _f64p, C.c_void_p, C.c_void_p]),
    "bcp_goal_direction_state": (C.c_int, [_H, _f64p, C.c_void_p, C.c_void_p]),
    "bcp_mini_world_seed": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bcp_sample_mini_worlds": (C.c_int, [_H, C.POINTER(BcpMiniWorldParams), C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcp_get_distance_field": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "bcp_get_near_field": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "bcp_plan_mini_worlds": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcp_refresh_mini_worlds": (C.c_int, [_H, C.POINTER(BcpMiniWorldParams)] + [C.c_void_p] * 6 + [C.c_double] +
                                [C.c_void_p] * 3),
    "bcp_release_mini_worlds": (C.c_int, [_H, C.c_void_p]),
    "bcp_mini_world_paths": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "bcp_device_normals": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "bcp_step_form": (C.c_int, [_H]),
    "bcp_time_step_kernels": (C.c_int, [_H, C.POINTER(BcpStepIO), C.c_uint32, C.c_int32, C.c_void_p,
                                        C.POINTER(C.c_float)]),
    "bcp_time_steps": (C.c_int, [_H, C.POINTER(BcpStepIO), C.c_uint32, C.c_int32, C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None


class BcpError(RuntimeError):
    """A libbcplan call returned a negative status."""


def load():
    """dlopen libbcplan.so and type every exported symbol.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbcplan.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc, gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError here means header and library disagree
            fn.restype = res
            fn.argtypes = args
        if L.bcp_abi_version() != ABI_VERSION:
            raise ImportError("libbcplan ABI %d != binding ABI %d" % (L.bcp_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise BcpError("libbcplan error %d: %s" % (rc, load().bcp_last_error().decode("utf-8", "replace")))
