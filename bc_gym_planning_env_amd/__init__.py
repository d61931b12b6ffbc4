"""bc_gym_planning_env_amd: MI355X-native batched PlanEnv.step() behind the reference's Observation / Action / State
API.  The compute lives in libbcplan.so (hand-written HIP for gfx950, C ABI in include/bcplan.h); this package is the
thin Python host side.  Importing the package never touches the GPU; creating an env without the built library or
without a GPU raises."""
from .api import (Action, Box, ContinuousRewardProviderState, CostMap2D, DiffdriveRobotState, EnvParams,  # noqa: F401
                  INDUSTRIAL_DIFFDRIVE_V1, INDUSTRIAL_TRICYCLE_V1, Observation, RewardParams, State,
                  TricycleRobotState)

__all__ = ["Action", "Box", "ContinuousRewardProviderState", "CostMap2D", "DiffdriveRobotState", "EnvParams",
           "INDUSTRIAL_DIFFDRIVE_V1", "INDUSTRIAL_TRICYCLE_V1", "Observation", "RewardParams", "State",
           "TricycleRobotState", "BatchedPlanEnv", "NativeOps"]


def __getattr__(name):
    # torch is imported lazily so that `import bc_gym_planning_env_amd` stays cheap for host-only users
    if name in ("BatchedPlanEnv", "BatchedState", "BatchedObservation"):
        from . import batched_env
        return getattr(batched_env, name)
    if name == "NativeOps":
        from .ops import NativeOps
        return NativeOps
    raise AttributeError(name)
