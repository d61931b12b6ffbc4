"""The C ABI on its own: a plain C11 program (tests/c_abi/step_from_c.c, gcc, no Python / torch / C++ in the process)
drives libbcplan.so through include/bcplan.h; its results must be those of the Python host layer on the same scene."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path, launcher):
    exe = str(tmp_path / "step_from_c")
    lib = os.path.join(ROOT, "bc_gym_planning_env_amd", "libbcplan.so")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "step_from_c.c"), lib, "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe]
    r = launcher.run(cmd, timeout=300)
    assert r["rc"] == 0, r["err"]
    return exe


def test_plain_c_client_matches_the_python_host_layer(torch_cuda, tmp_path, launcher):
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    exe = _compile(tmp_path, launcher)
    r = launcher.run([exe], timeout=120)
    assert r["rc"] == 0, r["err"]
    out = r["out"].splitlines()
    assert out[0].startswith("episodes_ended ")
    episodes = int(out[0].split()[1])
    rows = [line.split() for line in out[1:]]
    n = len(rows)
    assert n == 128
    c = {k: np.array([float.fromhex(r[j]) for r in rows]) for k, j in (("x", 1), ("y", 2), ("th", 3), ("rew", 6))}
    c_tgt = np.array([int(r[4]) for r in rows])
    c_it = np.array([int(r[5]) for r in rows])

    # the same scene through the Python host layer
    data = np.zeros((64, 64), dtype=np.uint8)
    data[8:56, 40] = 254
    path = np.stack([0.05 * np.arange(40), np.zeros(40), np.zeros(40)], axis=1)
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=0.05, refine_path=False,
                       iteration_timeout=45)
    env = BatchedPlanEnv(CostMap2D(data, 0.05, np.array([-0.5, -1.6])), path, params, n_envs=n, noise_parameters=None,
                         auto_reset=True)
    actions = np.stack([np.full(n, 0.5), 0.02 * (np.arange(n) % 21 - 10)], axis=1)
    total = torch.zeros(n, dtype=torch.float64, device="cuda")
    done_count = 0
    for _ in range(60):
        _, rew, done, _ = env.step(actions)
        total += rew
        done_count += int(done.sum())
    st = env.state.robot.cpu().numpy()
    assert episodes == done_count and episodes > n          # every env finished at least one episode (wall or timeout)
    assert (c["x"] == st[0]).all() and (c["y"] == st[1]).all() and (c["th"] == st[2]).all()
    assert (c_tgt == env.state.target_idx.cpu().numpy()).all() and (c_it == env.state.current_iter.cpu().numpy()).all()
    assert (c["rew"] == total.cpu().numpy()).all() and c["rew"].max() >= 5.0
