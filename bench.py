#!/usr/bin/env python
"""bench.py -- env-steps/s of the fused PlanEnv.step() path on RandomMiniEnv at 65 536 envs per GPU.

python bench.py --gpus N --steps K --warmup W

N > 1: either started once per rank by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- with WORLD_SIZE unset -- bench.py starts its N
ranks itself as fresh child processes BEFORE anything in the parent touches the GPU; the parent only relays rank 0's
JSON line and the exit codes.  One rank per GPU over RCCL; BCP_DIST_BACKEND=gloo is the explicit rehearsal transport
for boxes with fewer GPUs than ranks (never chosen silently: a failed RCCL set-up is an error).

Workload (BASELINE.json configs[2], "C3"): 65 536 replicas per GPU of the RandomMiniEnv seed-0 geometry (shared
183x183 uint8 costmap, shared refined path), tricycle dynamic model with PlanEnv's odometry noise drawn on the
device (Philox4x32-10), float32 actions ~ U(action_space) pre-staged in HBM, reset-on-done inside the kernel.
A "step" is one pass of the hot path over all envs of the rank.  With N > 1 ranks the env index space is sharded in
contiguous blocks (weak scaling); the done masks go into a device-side ring that is all-gathered once per 128-step rollout.

Timed region: barrier + synchronize, then `reps` x `--steps` steps back to back, then every rank drains ITS OWN stream
and gathers and stops its clock (the closing barrier comes after and is not timed); `reps` is chosen so that the region
lasts >= 50 ms (one 20-step block is 0.3 ms: shorter than a host synchronisation is exact).  value = envs x reps x steps
/ MAX over ranks of the wall time; the device time of the same region (HIP events) is printed beside it.
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
GATHER_EVERY = 128        # multi-GPU: the done masks of this many steps travel in one all-gather (see run_rank): one
                          # rollout of the reference's PPO runner (StepEnvRoller number_of_steps=128, scripts/rl_runners/ppo_runner.py:70)
MIN_REGION_MS = 50.0     # the timed region lasts at least this long (see run_rank: reps)
# ALGORITHMIC bytes per env-step (DESIGN.md "Bytes"): SoA state in + out (7 f64 robot + min_dist f64 +
# target_idx i32 + current_iter i32 + robot_collided u8 = 73 B each way) + action 2 x f32 + reward f64 + done u8.
# The shared costmap / path are LDS- and cache-resident and contribute no compulsory HBM traffic.
BYTES_PER_ENV_STEP = 73 + 73 + 8 + 8 + 1
BYTES_PER_ENV_STEP_C2 = 57 + 57 + 8 + 9            # diff-drive: 5 f64 robot state
BYTES_PER_ENV_STEP_C4 = BYTES_PER_ENV_STEP + 900 + 3120   # + footprint cells of a private map + private 130-point path
HBM_PEAK_GBS = 8000.0
METRIC = "env-steps/sec at N=65536 RandomMiniEnv, 1/2/4/8 MI355X; % HBM roofline"


# ------------------------------------------------------------------------------------------------ launcher (no GPU)
def spawn_ranks(args, script=None, argv=None):
    """Start one fresh process per rank and relay rank 0's output.  Nothing here imports torch or touches the GPU.
    (`script` / `argv`: the rank program, bench.py with this command line by default; the CPU test of the launcher
    runs its own rank program at world size 8.)"""
    n = args.gpus
    script = os.path.abspath(script or __file__)
    argv = sys.argv[1:] if argv is None else list(argv)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, cwd=os.getcwd()))
    deadline = time.time() + float(os.environ.get("BCP_BENCH_TIMEOUT", "1500"))
    rc = 0
    out0 = b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
            if rc != 0 or time.time() > deadline:
                if rc == 0:
                    rc = 124
                    sys.stderr.write("bench.py: ranks still running at the time limit; stopping them\n")
                break
            time.sleep(0.2)
    finally:
        for p in procs:   # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        out0 = procs[0].stdout.read() if procs[0].stdout else b""
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------ workloads
def make_env(n, device, env_id_base, seed, broadcast=False):
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_traj_mini_00.npz"))
    if broadcast:   # a sharded job: rank 0's copy of the shared costmap / origin / resolution / path for everybody
        from bc_gym_planning_env_amd import distributed as bdist
        names = ("costmap", "origin", "resolution", "path")
        g = dict(zip(names, bdist.broadcast_geometry([g[k] for k in names])))
    res = float(np.asarray(g["resolution"]).reshape(-1)[0])
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False)
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, device=device,
                         auto_reset=True, env_id_base=env_id_base, seed=seed)
    return env, g


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Cores this process may really use: the affinity mask, cut by the cgroup's CPU quota when there is one."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    avail = min(avail, max(1, int(int(fields[0]) / int(fields[1]))))
            else:
                quota = int(fields[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    period = int(f2.read().split()[0])
                if quota > 0:
                    avail = min(avail, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return avail


def cpu_baseline(g, envs=ENVS_PER_GPU, budget_s=12.0):
    """The oracle (C restatement, kind "port") on the host cores of this box, SURVEY 8(d): the C3 workload at the
    metric's N = 65 536, and the C1 case (one env, one core).  Every thread takes its block of envs through all the
    steps of a sample inside one library call (bco_run_steps): no Python and no thread start per step.  Bounded samples
    (a few seconds per thread count tried + about `budget_s` for the best one)."""
    import oracle
    avail = usable_cores()
    p = oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8)
    res = float(np.asarray(g["resolution"]).reshape(-1)[0])
    ref = oracle.OracleBatch(p, envs, g["costmap"], g["origin"], res, g["path"])
    ref.reset_from_paths()
    rng = np.random.RandomState(1)
    lo = np.array([np.pi / 30, -np.pi / 2])
    hi = np.array([np.pi / 6, np.pi / 2])
    acts = rng.uniform(lo, hi, (8, envs, 2)).astype(np.float32).astype(np.float64)
    zs = rng.standard_normal((8, envs, 3))
    ref.run_steps(acts, zs, 16, threads=min(avail, 16))  # de-synchronise the replicas as the GPU warm-up does

    def rate(threads, steps):
        t0 = time.perf_counter()
        ref.run_steps(acts, zs, steps, threads=threads)
        return envs * steps / (time.perf_counter() - t0)

    one = rate(1, 8)                       # about 2 s on one core
    tried = {}
    for t in sorted(set([min(avail, c) for c in (16, 32, 64, 128)])):   # the mask may be wider than what the box grants
        tried[t] = rate(t, max(8, int(2.0 * one * min(t, 16) / envs)))
    threads = max(tried, key=tried.get)
    steps = max(16, int(budget_s * tried[threads] / envs))
    t0 = time.perf_counter()
    ref.run_steps(acts, zs, steps, threads=threads)
    dt = time.perf_counter() - t0
    # C1 (BASELINE.json configs[0]): ONE RandomMiniEnv, random actions, reset on done -- the reference's own shape
    ref1 = oracle.OracleBatch(p, 1, g["costmap"], g["origin"], res, g["path"])
    ref1.reset_from_paths()
    a1 = rng.uniform(lo, hi, (4096, 1, 2)).astype(np.float32).astype(np.float64)
    z1 = rng.standard_normal((4096, 1, 3))
    c1_steps = max(4096, int(3.0 * one))
    t2 = time.perf_counter()
    ref1.run_steps(a1, z1, c1_steps, threads=1)
    dt2 = time.perf_counter() - t2
    return {"value": envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "cores_usable": avail,
            "sample": "C3 at N = %d: %d steps of the same workload (C oracle restatement, %d threads, %.1f s)"
                      % (envs, steps, threads, dt),
            "thread_counts_tried": dict((str(k), v) for k, v in tried.items()),
            "single_core_value": one,
            "c1_single_env": {"value": c1_steps / dt2, "unit": "env-steps/s", "cores": 1,
                              "sample": "C1: one RandomMiniEnv, %d steps, reset on done (%.1f s)" % (c1_steps, dt2)},
            "reference_python": "genuine reference, 1 core of the build container: about 2.3e3 env-steps/s (BASELINE.md; "
                                "it cannot travel to the GPU box)"}


def newest_profile(name):
    """Newest committed profiles/rNN_<name> (by round number, exact name): (path, parsed json) or (None, None)."""
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_" + name)):
        m = re.match(r"^r(\d+)_" + re.escape(name) + "$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, None
    try:
        with open(best[1]) as f:
            data = json.load(f)
        data["corrected_bytes_per_step"]["total"]   # (the field the bench line quotes)
        return os.path.relpath(best[1], ROOT), data
    except (OSError, ValueError, KeyError, TypeError):
        return None, None


def issue_object(name="step_alu_pmc.json", waves_per_simd=4):
    """The resource that binds the shared-map step (ALU / latency, not HBM): what the SQ counters of the newest
    committed profiles/rNN_<name> say about the step kernel's waves.  PMC passes cannot run inside bench.py; the file is named."""
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_" + name)):
        m = re.match(r"^r(\d+)_" + re.escape(name) + "$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None
    try:
        with open(best[1]) as f:
            d = json.load(f)["derived"]["step_local_kernel"]
        valu = d["valu_active_fraction_of_wave_cycles"]
        return {"bound": "instruction issue + dependent-chain latency (no HBM or MFMA roofline applies: 163 B and no contraction per env-step)",
                "valu_busy_per_simd": valu * waves_per_simd, "waves_per_simd": waves_per_simd,
                "wave_cycles": {"waiting": d["waiting_fraction_of_wave_cycles"], "issuing": d["issuing_fraction_of_wave_cycles"],
                                "issue_stalled": d["issue_stalled_fraction_of_wave_cycles"], "valu_active": valu,
                                "scalar_active": d.get("scalar_active_fraction_of_wave_cycles"),
                                "lds_active": d.get("lds_active_fraction_of_wave_cycles")},
                "instructions_per_wave": {k.replace("_instructions_per_wave", ""): v for k, v in d.items() if k.endswith("_instructions_per_wave")},
                "icache_miss_fraction": d.get("icache_miss_fraction_of_requests"),
                "what": "valu_busy_per_simd = fraction of its lifetime in which a wave has a VALU instruction executing x the waves "
                        "resident per SIMD: the share of the step during which a SIMD's vector ALU is busy (1 = saturated)",
                "source": os.path.relpath(best[1], ROOT)}
    except (OSError, ValueError, KeyError, TypeError):
        return None


def parked_fraction(env, run, steps):
    """Share of env-steps whose pose needed the exact footprint test (bcp_parked_poses) over `steps` steps driven by run()."""
    before = env.parked_poses()
    run()
    return (env.parked_poses() - before) / float(env.n_envs * steps)


def steady_state(env, pool, rng):
    """Pre-roll to the steady state of the rollout: random episode phases, then one full timeout's worth of steps."""
    import torch
    n = env.n_envs
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, env.params.iteration_timeout, n).astype(np.int32)).to(env.device))
    for k in range(env.params.iteration_timeout):
        env.step(pool[k % pool.shape[0]])
    torch.cuda.synchronize()


def make_c4_env(n, device, seed=11):
    """BASELINE.json configs[3]: n AisleTurnEnv replicas (10 m / 256 px, the four flip variants) with PRIVATE costmaps stored
    uint8 [n, 256, 256] (valid 256 x 141) and private 130-point paths, tricycle + PlanEnv noise, reset on done."""
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    G = os.path.join(ROOT, "tests", "golden")
    names = ["g8_traj_aisle_c4_00.npz", "g8_traj_aisle_c4_10.npz", "g8_traj_aisle_c4_01.npz", "g8_traj_aisle_c4_11.npz"]
    gs = [np.load(os.path.join(G, nm)) for nm in names]
    res = float(gs[0]["resolution"])
    params = EnvParams(resolution=res, refine_path=False)
    cms = [CostMap2D(x["costmap"], res, x["origin"]) for x in gs]
    return BatchedPlanEnv(cms, [x["path"] for x in gs], params, n_envs=n, auto_reset=True, template_of_env=np.arange(n) % 4,
                          map_storage=(256, 256), device=device, seed=seed)


def aux_other_configs(device):
    """BASELINE.json configs[1] (C2) and configs[3] (C4), informational lines with their own roofline objects."""
    import torch
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    G = os.path.join(ROOT, "tests", "golden")
    out = {}
    rng = np.random.RandomState(0)
    # ---- C2: 4096 diff-drive envs, shared 64x64 costmap, noise off
    g = np.load(os.path.join(G, "g8dd_traj_mini64_00.npz"))
    res = float(g["resolution"])
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False,
                       robot_name="industrial_diffdrive_v1")
    n = 4096
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, noise_parameters=None,
                         auto_reset=True, device=device)
    pool = torch.from_numpy(np.stack([rng.uniform([0.105, -np.pi / 2], [0.524, np.pi / 2], (n, 2)).astype(np.float32)
                                      for _ in range(8)])).to(env.device)
    steady_state(env, pool, rng)
    ms = env.time_steps(pool[0], 200)
    parked_c2 = parked_fraction(env, lambda: env.time_steps(pool[0], 100), 100)
    out["c2_diffdrive_4096_shared_64x64"] = {
        "what": "BASELINE configs[1]: RandomMiniEnv geometry at 5.5 m / 64 px, 4096 envs, diff-drive model, noise off",
        "ms_per_step": ms, "env_steps_per_s": n / (ms * 1e-3), "parked_pose_fraction": parked_c2,
        "roofline": {"bound": "hbm", "achieved": BYTES_PER_ENV_STEP_C2 * n / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": BYTES_PER_ENV_STEP_C2 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP_C2 * n,
                     "note": "64 wavefronts of envs: launch-latency bound, the chip is idle"}}
    env.close()
    del env, pool
    # ---- C4: 65 536 AisleTurn envs, PRIVATE costmaps stored [N, 256, 256] (valid 256 x 141) and private 130-point paths
    n = ENVS_PER_GPU
    t0 = time.perf_counter()
    env = make_c4_env(n, device)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).to(env.device)
    steady_state(env, pool, rng)
    ms = env.time_steps(pool[0], 100)
    parked = parked_fraction(env, lambda: env.time_steps(pool[0], 50), 50)
    path, pmc = newest_profile("c4_pmc_summary.json")
    traffic = pmc["corrected_bytes_per_step"]["total"] if pmc else None
    ach = (traffic / (ms * 1e-3) / 1e9) if traffic else None
    survey = BYTES_PER_ENV_STEP_C4 * n / (ms * 1e-3) / 1e9
    # what re-binding the 65 536 private maps costs (bcp_set_costmaps: lethal bitmaps, distance fields -- edt_lds_kernel --
    # and their 1-bit tiles): irrelevant for static maps, the price of a RandomAisleTurnEnv that redraws its map on reset
    k = env._keep
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(env.device))
    env.set_costmap_tensors(k["map"], k["origins"], env.resolution, k["vr"], k["vc"])
    e1.record(torch.cuda.current_stream(env.device))
    torch.cuda.synchronize()
    rebuild_ms = e0.elapsed_time(e1)
    setup = {"derived_map_data_rebuild_ms": rebuild_ms, "per_map_us": rebuild_ms * 1e3 / n,
             "what": "bcp_set_costmaps on the resident [N, 256, 256] maps: pack_bitmap_kernel + near_dilate_kernel (1-bit tiles straight from "
                     "the lethal masks; the uint8 fields follow on demand: round 3 built them here with edt_lds_kernel, 75 ms)"}
    out["c4_aisle_private_maps"] = {
        "what": "BASELINE configs[3]: AisleTurnEnv at 10 m / 256 px, 65536 envs, private costmaps stored uint8 "
                "[N, 256, 256] (valid 256 x 141, 4 templates x flips) + private 130-point paths, tricycle + noise",
        "map_storage": list(env._keep["map"].shape), "setup_s": t_setup, "setup_detail": setup,
        "ms_per_step": ms, "env_steps_per_s": n / (ms * 1e-3),
        "collisions_per_step": float(env.collided_now.float().mean()), "done_per_step": float(env.done.float().mean()),
        "parked_pose_fraction": parked,
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (ach / HBM_PEAK_GBS) if ach else None,
                     "traffic": traffic, "traffic_source": path,
                     "bytes_basis": "HBM bytes the kernel really moves per launch (FETCH_SIZE / WRITE_SIZE passes of the committed "
                                    "profile, calibrated) / this run's launch time: <= 1 by construction",
                     "survey_algorithmic": {"bytes_per_launch": BYTES_PER_ENV_STEP_C4 * n, "gbs_if_moved": survey,
                                            "quotient_of_peak": survey / HBM_PEAK_GBS, "not_a_utilisation": True,
                                            "note": "SURVEY 8(d)'s 163 B state I/O + 900 footprint cells of the private uint8 map + "
                                                    "3120 B of private path per env-step; the kernel reads 1-bit lethal masks, 1-bit "
                                                    "distance tiles, 8-byte quantised prefilter records and a bucketed path window instead and "
                                                    "does not move these bytes"}}}
    env.close()
    return out


def aux_ego_aisle(device, n=ENVS_PER_GPU):
    """The observation the reference's own vectorised consumer builds (scripts/rl_runners/ppo_runner.py:35-43:
    ColoredEgoCostmapRandomAisleTurnEnv, envs/synth_turn_env.py:376-451) and EgocentricCostmap on the default AisleTurn map:
    n replicas of the fixture's geometry at steady state, one observation call per timed repetition."""
    import torch
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    from bc_gym_planning_env_amd.egocentric import BatchedColoredEgoCostmap, BatchedEgocentricCostmap
    out = {}
    rng = np.random.RandomState(5)
    for key, fixture, wrapper, what in (
            ("egocentric_aisle_default_map", "g10_ego_aisle.npz", BatchedEgocentricCostmap,
             "EgocentricCostmap(AisleTurnEnv()) (envs/egocentric.py:102-160)"),
            ("colored_ego_aisle_350x512", "g12_colored_ego.npz", BatchedColoredEgoCostmap,
             "ColoredEgoCostmapRandomAisleTurnEnv's observation (envs/synth_turn_env.py:376-451), the env of ppo_runner.py:35-43")):
        g = np.load(os.path.join(ROOT, "tests", "golden", fixture))
        res = float(g["resolution"])
        env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], EnvParams(resolution=res, refine_path=False),
                             n_envs=n, auto_reset=True, device=device, seed=17)
        pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).to(env.device)
        steady_state(env, pool, rng)
        wrap = wrapper(env)
        stream = torch.cuda.current_stream(env.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            wrap.observation()
        reps = 20
        e0.record(stream)
        for _ in range(reps):
            wrap.observation()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        route = wrap.route()
        img_bytes = wrap.images.numel()
        env.set_tuning(ego_sparse=0)          # the sampling kernel the same call fell to before round 4
        for _ in range(2):
            wrap.observation()
        e0.record(stream)
        for _ in range(5):
            wrap.observation()
        e1.record(stream)
        torch.cuda.synchronize()
        ms_dense = e0.elapsed_time(e1) / 5
        dense_route = wrap.route()
        env.set_tuning(ego_sparse=1)
        ach = img_bytes / (ms * 1e-3) / 1e9
        out[key] = {"what": "%s for %d envs: costmap %s -> %d x %d px uint8 per env" % ((what, n, list(g["costmap"].shape)) + wrap.image_shape),
                    "kernel": route["kernel"], "non_zero_cells_of_the_map": route["max_cells"], "sparse_limit": route["limit"],
                    "ms_per_call": ms, "bytes_written_per_call": img_bytes,
                    "lethal_pixel_fraction": float((wrap.images == 254).float().mean()),
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "bytes_basis": "algorithmic = bytes written: every image byte is stored once, inputs are cache-resident"},
                    "sampling_kernel_same_call": {"kernel": dense_route["kernel"], "ms_per_call": ms_dense,
                                                  "frac": img_bytes / (ms_dense * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        env.close()
        del wrap, env, pool
        torch.cuda.empty_cache()
    return out


def aux_measurements(env, pool, n):
    """Informational, rank 0 at N=1 only (NOT the metric): the neighbours of the step path built per SURVEY 8(f),
    measured on the same steady-state batch with HIP events on the launch stream."""
    import torch
    from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap
    out = {}
    stream = torch.cuda.current_stream(env.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wrap = BatchedEgocentricCostmap(env)
    for k in range(3):
        wrap.observation()
    reps = 20
    e0.record(stream)
    for k in range(reps):
        wrap.observation()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    img_bytes = wrap.images.numel()
    e0.record(stream)
    for k in range(reps):
        wrap.step(pool[k % 16])
    e1.record(stream)
    torch.cuda.synchronize()
    ms_both = e0.elapsed_time(e1) / reps
    out["egocentric_observation"] = {
        "what": "EgocentricCostmap.observation for every env: %d x %d px uint8 + goal_n_state (envs/egocentric.py:102-160)"
                % wrap.image_shape,
        "kernel": wrap.route()["kernel"], "non_zero_cells_of_the_map": wrap.route()["max_cells"], "ms_per_call": ms, "bytes_written_per_call": img_bytes,
        "roofline": {"bound": "hbm", "achieved": img_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": img_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "step_plus_observation_ms": ms_both, "env_steps_per_s_with_observation": n / (ms_both * 1e-3)}
    del wrap
    # K steps per launch for callers that hold a whole rollout's actions (bcp_rollout): NOT the metric -- the metric is
    # one launch per step, PlanEnv.step's contract
    k_roll = 128
    acts = pool[torch.arange(k_roll, device=pool.device) % 16].contiguous()
    for _ in range(2):
        env.rollout(acts)
    e0.record(stream)
    for _ in range(8):
        env.rollout(acts)
    e1.record(stream)
    torch.cuda.synchronize()
    ms_roll = e0.elapsed_time(e1) / (8 * k_roll)
    out["rollout_128_steps_per_launch"] = {
        "what": "bcp_rollout: %d steps of the metric workload per call = ONE launch of step_local_kernel<.., ROLL>, the workgroups "
                "advancing independently (open-loop Monte-Carlo rollouts, the reference's README use case); bit for bit %d calls "
                "of bcp_step (tests/test_gpu_rollout.py)" % (k_roll, k_roll),
        "ms_per_step": ms_roll, "env_steps_per_s": n / (ms_roll * 1e-3)}
    del acts
    # a fresh world per episode: RandomMiniEnv.reset() as a device-side walk through a pool of pre-sampled geometries
    from bc_gym_planning_env_amd import mini_env
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    worlds = mini_env.sample_pool_device(None, list(range(4096)), 4, device=env.device.index or 0)
    t_pool = time.perf_counter() - t0
    penv = mini_env.BatchedRandomMiniEnv(n, pool=worlds, auto_reset=True, seed=3, device=env.device.index or 0)
    rng = np.random.RandomState(7)
    penv.state.current_iter.copy_(torch.from_numpy(rng.randint(0, penv.params.iteration_timeout, n).astype(np.int32)).to(env.device))
    for k in range(penv.params.iteration_timeout):
        penv.step(pool[k % 16])
    torch.cuda.synchronize()
    e0.record(stream)
    for k in range(100):
        penv.step(pool[k % 16])
    e1.record(stream)
    torch.cuda.synchronize()
    ms_pool = e0.elapsed_time(e1) / 100
    out["geometry_pool"] = {
        "what": "RandomMiniEnv with draw_new_turn_on_reset: %d envs over %d pre-sampled worlds (%d chains x %d), every "
                "reset moves the env to its chain's next world inside the step kernel" % (n, len(worlds), 4096, 4),
        "ms_per_step": ms_pool, "env_steps_per_s": n / (ms_pool * 1e-3),
        "episodes_ending_per_step": float(penv.done.float().sum()),
        "device_sampling_worlds_per_s": len(worlds) / t_pool,
        "sampler": "bcp_sample_mini_worlds: numpy's MT19937 stream + rejection sampler + walls + acceptance test, one "
                   "wavefront per stream (rate includes the download and the host-side pool objects)"}
    del penv, worlds
    # ... and worlds that never repeat: one RandomState stream per env, the pool entries of a stream are a ring that is
    # re-sampled behind the env on a side stream while the steps go on (plan / refresh / release)
    eenv = mini_env.BatchedRandomMiniEnv(n, episodes=8, endless=True, auto_reset=True, seed=3, device=env.device.index or 0)
    eenv.state.current_iter.copy_(torch.from_numpy(rng.randint(0, eenv.params.iteration_timeout, n).astype(np.int32)).to(env.device))
    infos = []
    for k in range(256):
        eenv.step(pool[k % 16])
        if k % 128 == 127:
            eenv.refresh(overlap=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for k in range(512):
        eenv.step(pool[k % 16])
        if k % 128 == 127:
            info = eenv.refresh(overlap=True)
            if info is not None:
                infos.append(info.clone())
    e1.record(stream)
    torch.cuda.synchronize()
    ms_endless = e0.elapsed_time(e1) / 512
    tally = torch.stack(infos).sum(0).cpu().numpy() if infos else np.zeros(4)
    out["endless_geometry_pool"] = {
        "what": "RandomMiniEnv(seed=i) for every env i, for ever: %d streams x 8 ring entries, refresh(overlap=True) every "
                "128 steps re-samples the worlds the envs have left (MT19937 stream order) on a side stream" % n,
        "ms_per_step": ms_endless, "env_steps_per_s": n / (ms_endless * 1e-3),
        "worlds_resampled_per_refresh": float(tally[0]) / max(len(infos), 1),
        "envs_seen_waiting_for_worlds": int(tally[1])}
    del eenv
    return out


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch
    import torch.distributed as dist
    from bc_gym_planning_env_amd import distributed as bdist

    try:
        rank, world, local_rank = bdist.init_from_env()
    except Exception as exc:   # a failed RCCL / rendezvous set-up is an error, never a silent change of transport
        sys.stderr.write("bench.py: torch.distributed set-up failed: %s\n" % (exc,))
        raise SystemExit(3)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    device = bdist.local_device(local_rank)
    torch.cuda.set_device(device)
    n = args.envs_per_gpu
    sharded = dist.is_initialized()     # world > 1, or a forced group at world size 1 (BCP_DIST_FORCE=1)
    # the replicated geometry comes from rank 0 (one broadcast at set-up), and every rank proves it holds the same
    env, g = make_env(n, device, env_id_base=rank * n, seed=2024, broadcast=sharded)
    bdist.check_same_geometry(env.geometry_digest())
    backend = dist.get_backend() if sharded else None
    # Multi-GPU: the only cross-rank traffic is the done mask.  Every rank writes its mask of step k into row k % R of a
    # ring (the step kernel stores it there directly) and the ring is all-gathered every R steps, asynchronously (the
    # gather of one block overlaps the kernels of the next).  R = 128: the consumer the reference itself has, its PPO
    # runner, collects 128 steps per rollout before it looks at anything (ppo_runner.py:70).  One gather costs ~31 us of
    # launch work on the host and of stream time whatever it carries (measured with the RCCL branch forced at world size 1:
    # R = 8 15.4 us per step, 32 12.6, 128 11.9, no group at all 11.55), so a short ring would measure the collective's
    # launch, not the step.  The rows of an unfinished block are gathered at the end of the timed region, inside it.
    # A sharded run also times rounds 1-2's definition (R = 8) in a second region and prints it beside the metric
    # ("gather_every_8"), so that lines of different rounds can be compared.
    # pre-staged synthetic actions: a pool of 16 batches ~ U(action_space), float32, resident in HBM
    rng = np.random.RandomState(1234 + rank)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).to(env.device)
    stream = torch.cuda.current_stream(env.device)  # the stream libbcplan launches on

    def max_over_ranks(x):
        if not sharded:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=env.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    state = {"k0": 0, "pre_rolled": False}

    def measure(gather_every, warmup):
        """One timed region with the done-mask ring of `gather_every` rows: dict(elapsed, stream_ms, total, reps)."""
        ring = torch.zeros(gather_every, n, dtype=torch.uint8, device=torch.device("cuda", device)) if sharded else None
        gather = bdist.DoneGather(gather_every * n, torch.device("cuda", device), packed=True) if sharded else None   # (one bit per env on the links)
        if sharded:
            # the first collectives run here, untimed (communicator set-up)
            gather.launch(ring.view(-1))
            gather.flush()
            dist.barrier()
            torch.cuda.synchronize()

        def one_step(k):
            if gather is None:
                env.step(pool[k % 16])
                return
            env.step(pool[k % 16], done_out=ring[k % gather_every])
            if k % gather_every == gather_every - 1:
                gather.launch(ring.view(-1))

        def drain():
            """This rank's own work is finished: its gathers have landed and its stream is empty.  No collective in here."""
            if gather is not None:
                gather.flush()
            torch.cuda.synchronize()

        def barrier():
            drain()
            if sharded:
                dist.barrier()
                torch.cuda.synchronize()

        # Pre-roll to the steady state of the rollout (untimed set-up): every env gets a random episode phase, then one
        # full timeout's worth of steps runs, so that at any timed step the batch holds envs at all stages of an
        # episode (fresh, en route, off the map, about to time out) instead of 65 536 envs marching in lock-step.
        if not state["pre_rolled"]:
            steady_state(env, pool, rng)
            state["pre_rolled"] = True
        if gather is not None:   # first collectives outside the timed region whatever --warmup says (communicator set-up)
            for _ in range(2):
                gather.launch(ring.view(-1))
        barrier()

        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0 = state["k0"]
        for k in range(warmup):
            one_step(k0 + k)
        k0 += warmup
        # How long is one block of `steps` steps?  (an untimed probe, so that --warmup 0 works too.)  A block shorter
        # than MIN_REGION_MS is no instrument -- at the driver's --steps 20 it lasts 0.3 ms and one host synchronisation is
        # 9 % of it -- so the timed region repeats the block `reps` times back to back; `steps` stays what was asked for,
        # ms_per_step and value are per step.  Every rank uses the same `reps` (MAX of the probes).
        # (the probe's first block only fills the launch queue -- timed from a standing start it reads up to 20 % slow, and
        #  the region then ends before MIN_REGION_MS --, its second block is the one that is timed; 10 % on top)
        probe = max(1, min(args.steps, 32))
        barrier()
        for k in range(probe):
            one_step(k0 + k)
        ev0.record(stream)
        for k in range(probe):
            one_step(k0 + probe + k)
        ev1.record(stream)
        k0 += 2 * probe
        drain()
        probe_ms = max_over_ranks(ev0.elapsed_time(ev1) / probe)
        reps = args.reps if args.reps > 0 else int(min(8192, max(1, np.ceil(1.1 * MIN_REGION_MS / max(probe_ms * args.steps, 1e-6)))))
        k0 = (k0 + gather_every - 1) // gather_every * gather_every   # (the ring starts the region at row 0)
        total = reps * args.steps

        # ---- the timed region: barrier + synchronize | reps x steps steps | this rank's stream and gathers drained ----
        # The clock of a rank stops when ITS work is done (drain: no collective); the closing barrier comes after it and
        # is not part of what `value` is computed from; the job's time is the MAX over ranks.  Beside the wall clock the
        # device time of the same region (HIP events on the launch stream, MAX over ranks).
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(total):
            one_step(k0 + k)
        if gather is not None and total % gather_every:   # (the rows of the last, unfinished block)
            gather.launch(ring.view(-1))
        ev1.record(stream)
        drain()
        elapsed_local = time.perf_counter() - t0
        barrier()
        stream_ms_local = ev0.elapsed_time(ev1)
        state["k0"] = k0 + total
        return {"elapsed": max_over_ranks(elapsed_local), "stream_ms": max_over_ranks(stream_ms_local), "total": total, "reps": reps}

    gather_every = max(1, int(args.gather_every))
    main_run = measure(gather_every, args.warmup)
    short_run = measure(8, 0) if sharded and gather_every != 8 else None   # rounds 1-2's ring, for comparison across rounds
    elapsed, stream_ms, total, reps = main_run["elapsed"], main_run["stream_ms"], main_run["total"], main_run["reps"]
    env.check_errors()

    # The step's launches run back to back on one stream; their combined average duration is measured live with HIP
    # events recorded on the launch stream around the timed region (at N=1 the region holds nothing but these
    # launches).  The per-kernel split of the same command is in profiles/ (rocprofv3 --kernel-trace --stats).
    step_ms = stream_ms / total
    achieved = BYTES_PER_ENV_STEP * n / (step_ms * 1e-3) / 1e9
    traffic = traffic_src = None
    if n == ENVS_PER_GPU:   # PMC passes cannot run inside bench.py; the newest committed summary is quoted and named
        traffic_src, pmc = newest_profile("pmc_summary.json")
        if pmc:
            traffic = pmc["corrected_bytes_per_step"]["total"]

    if rank == 0:
        total_envs = n * world
        kernels = env.step_kernels()
        out = {
            "metric": METRIC,
            "value": total_envs * total / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / total * 1e3,
            "reps": reps,
            "timed_region": {"steps_timed": total, "wall_ms": elapsed * 1e3, "device_ms": stream_ms,
                             "device_ms_per_step": stream_ms / total,
                             "what": "reps x steps steps back to back between barrier + synchronize and the drain of "
                                     "every rank's own stream and gathers; wall = MAX over ranks of the host clock, "
                                     "device = MAX over ranks of HIP events on the launch stream; the closing barrier "
                                     "is outside both"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C3: RandomMiniEnv seed-0 geometry, %d envs/GPU, tricycle dynamic model + PlanEnv "
                                   "odometry noise (on-device Philox), shared 183x183 costmap, reset on done, steady-state episode phases" % n,
                       "envs_total": total_envs, "envs_per_gpu": n, "actions": "float32 U(action_space), pre-staged",
                       "sharding": ("env blocks per rank, done masks ring-buffered on the device and all-gathered as bits (%s) every "
                                    "%d steps, overlapped with the next steps"
                                    % ("RCCL" if backend == "nccl" else "gloo through the host: REHEARSAL transport, not a scaling number", gather_every))
                       if sharded else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernels, "kernel_ms": step_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * n,
                         "traffic_source": traffic_src,
                         "issue": issue_object(),
                         "note": "shared-map config is latency-bound (SURVEY 8d): the HBM fraction is low by "
                                 "construction; see DESIGN.md.  The HBM-bound config is aux.c4_aisle_private_maps"},
        }
        out["gather_every"] = gather_every if sharded else None
        out["done_masks_packed"] = bool(sharded)
        if short_run is not None:
            out["gather_every_8"] = {"value": total_envs * short_run["total"] / short_run["elapsed"], "unit": "env-steps/s",
                                     "ms_per_step": short_run["elapsed"] / short_run["total"] * 1e3,
                                     "device_ms_per_step": short_run["stream_ms"] / short_run["total"],
                                     "what": "the same workload with the done-mask ring of rounds 1-2 (one all-gather per 8 steps, "
                                             "bit-packed): a second timed region of this run, NOT the metric"}
        out["parked_pose_fraction"] = parked_fraction(env, lambda: [env.step(pool[k % 16]) for k in range(64)], 64)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g)
        if world == 1 and not args.no_aux:
            try:  # informational only: never let it cost the metric line
                # (the observation legs on the AisleTurn maps come first: measured behind the pool legs -- tens of GB of
                #  buffers allocated and freed -- the 1.16 GB image buffer of the 133 x 133 window was written 15 x slower,
                #  3.4 - 4.0 ms per call in three runs against 0.265 - 0.27 ms in every run without that history, whatever
                #  the kernel version: the mapping of a fresh allocation, not the kernel)
                aux = aux_ego_aisle(device)
                aux.update(aux_measurements(env, pool, n))
                env.close()
                del env
                torch.cuda.empty_cache()
                aux.update(aux_other_configs(device))
                out["aux"] = aux
            except Exception as exc:  # noqa: BLE001
                out["aux"] = {"error": repr(exc)}
        print(json.dumps(out))
        sys.stdout.flush()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--reps", type=int, default=0, help="repeat the block of --steps steps this many times inside the "
                    "timed region (default: as many as make the region last %.0f ms)" % MIN_REGION_MS)
    ap.add_argument("--gather-every", type=int, default=GATHER_EVERY, help="multi-GPU: steps between two all-gathers of the done-mask ring")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the informational legs (observation, pools, C2 / C4)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
