#!/usr/bin/env python
"""bench.py -- env-steps/s of the fused PlanEnv.step() kernel on RandomMiniEnv at 65 536 envs per GPU.

python bench.py --gpus N --steps K --warmup W     (N > 1: launched through torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2], "C3"): 65 536 replicas per GPU of the RandomMiniEnv seed-0 geometry (shared
183x183 uint8 costmap, shared refined path), tricycle dynamic model with PlanEnv's odometry noise drawn on the
device (Philox4x32-10), float32 actions ~ U(action_space) pre-staged in HBM, reset-on-done inside the kernel.
A "step" is one fused kernel launch over all envs of the rank.  With N > 1 ranks the env index space is sharded in
contiguous blocks (weak scaling) and each step ends with ONE RCCL all-gather of the uint8 done mask.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
# ALGORITHMIC bytes per env-step (DESIGN.md "Bytes"): SoA state in + out (7 f64 robot + min_dist f64 + target_idx
# i32 + current_iter i32 + robot_collided u8 = 73 B each way) + action 2 x f32 + reward f64 + done u8.
# The shared costmap / path are LDS- and cache-resident and contribute no compulsory HBM traffic.
BYTES_PER_ENV_STEP = 73 + 73 + 8 + 8 + 1
HBM_PEAK_GBS = 8000.0


def make_env(n, device, env_id_base, seed):
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_traj_mini_00.npz"))
    res = float(g["resolution"])
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False)
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, device=device,
                         auto_reset=True, env_id_base=env_id_base, seed=seed)
    return env, g


def cpu_baseline(g, envs=16384, budget_s=12.0):
    """The oracle (C restatement, kind "port") on the host cores of this box: same workload, bounded sample
    (about `budget_s` seconds of wall time on at most 16 threads, the CPU share of a one-GPU box)."""
    import oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(16, avail))
    p = oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8)
    ref = oracle.OracleBatch(p, envs, g["costmap"], g["origin"], float(g["resolution"]), g["path"])
    ref.reset_from_paths()
    rng = np.random.RandomState(1)
    lo = np.array([np.pi / 30, -np.pi / 2])
    hi = np.array([np.pi / 6, np.pi / 2])
    acts = [rng.uniform(lo, hi, (envs, 2)).astype(np.float32).astype(np.float64) for _ in range(4)]
    zs = [rng.standard_normal((envs, 3)) for _ in range(4)]
    for k in range(20):  # de-synchronise the replicas as the GPU warm-up does
        ref.step(acts[k % 4], zs[k % 4], auto_reset=True, threads=threads)
    steps = 0
    t0 = time.perf_counter()
    while True:
        for k in range(10):
            ref.step(acts[(steps + k) % 4], zs[(steps + k) % 4], auto_reset=True, threads=threads)
        steps += 10
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= 2000:
            break
    # the same on ONE core, for a per-core figure (about 3 s)
    one = 0
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 3.0:
        ref.step(acts[one % 4], zs[one % 4], auto_reset=True, threads=1)
        one += 1
    dt1 = time.perf_counter() - t1
    return {"value": envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d envs x %d steps of the same workload (C oracle restatement, %d threads, %.1f s)"
                      % (envs, steps, threads, dt),
            "single_core_value": envs * one / dt1}


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
        "kernel": "ego_costmap_kernel", "ms_per_call": ms, "bytes_written_per_call": img_bytes,
        "roofline": {"bound": "hbm", "achieved": img_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": img_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "step_plus_observation_ms": ms_both, "env_steps_per_s_with_observation": n / (ms_both * 1e-3)}
    del wrap
    # a fresh world per episode: RandomMiniEnv.reset() as a device-side walk through a pool of pre-sampled geometries
    import time as _time
    from bc_gym_planning_env_amd import mini_env
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    worlds = mini_env.sample_pool_device(None, list(range(4096)), 4, device=env.device.index or 0)
    t_pool = _time.perf_counter() - t0
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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the informational egocentric-observation timing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bc_gym_planning_env_amd import distributed as bdist

    rank, world, local_rank = bdist.init_from_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    device = local_rank % max(1, torch.cuda.device_count())  # (rehearsals may put several ranks on one GPU)
    torch.cuda.set_device(device)
    n = args.envs_per_gpu
    env, g = make_env(n, device, env_id_base=rank * n, seed=2024)
    # Multi-GPU: the only cross-rank traffic is the done mask.  Every rank writes its mask of step k into row k % 8 of a
    # ring (the step kernel stores it there directly) and the ring is all-gathered every 8 steps, asynchronously
    # (the gather of one block of 8 steps overlaps the kernels of the next): 1/8 collective per step.
    gather_every = 8
    ring = torch.zeros(gather_every, n, dtype=torch.uint8, device=torch.device("cuda", device)) if world > 1 else None
    gather = bdist.DoneGather(gather_every * n, torch.device("cuda", device)) if world > 1 else None
    collective = None
    if world > 1:
        # the first collectives run here, untimed (communicator set-up)
        gather.launch(ring.view(-1))
        gather.flush()
        dist.barrier()
        torch.cuda.synchronize()
        collective = dist.get_backend()

    # pre-staged synthetic actions: a pool of 16 batches ~ U(action_space), float32, resident in HBM
    rng = np.random.RandomState(1234 + rank)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).to(env.device)

    def one_step(k):
        if gather is None:
            env.step(pool[k % 16])
            return
        env.step(pool[k % 16], done_out=ring[k % gather_every])
        if k % gather_every == gather_every - 1:
            gather.launch(ring.view(-1))

    def barrier():
        if world > 1:
            gather.flush()
            dist.barrier()
        torch.cuda.synchronize()

    # Pre-roll to the steady state of the rollout (untimed set-up): every env gets a random episode phase, then one
    # full timeout's worth of steps runs, so that at any timed step the batch holds envs at all stages of an
    # episode (fresh, en route, off the map, about to time out) instead of 65 536 envs marching in lock-step.
    phase = torch.from_numpy(rng.randint(0, env.params.iteration_timeout, n).astype(np.int32)).to(env.device)
    env.state.current_iter.copy_(phase)
    for k in range(env.params.iteration_timeout):
        env.step(pool[k % 16])
    if gather is not None:   # first collectives outside the timed region whatever --warmup says (communicator set-up)
        for _ in range(2):
            gather.launch(ring.view(-1))
    barrier()

    for k in range(args.warmup):
        one_step(k)
    barrier()
    stream = torch.cuda.current_stream(env.device)  # the stream libbcplan launches on
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        one_step(args.warmup + k)
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    stream_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=env.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    env.check_errors()

    # One step = two launches on one stream: step_fast_pair_kernel (all envs: robot model, O(1) collision classification,
    # reward, done) and step_pending_kernel (the ~1.5 % of envs whose collision needs the exact rasteriser).  Their
    # combined average duration is measured live with HIP events recorded on the launch stream around the timed
    # region (at N=1 the region holds nothing but these launches, back to back); the per-kernel split of the same
    # command is in profiles/ (rocprofv3 --kernel-trace --stats).
    step_ms = stream_ms / args.steps if world == 1 else env.time_steps(pool[0], max(20, min(args.steps, 200)))
    achieved = BYTES_PER_ENV_STEP * n / (step_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if os.path.exists(pmc) and n == ENVS_PER_GPU:  # PMC passes cannot run inside bench.py; committed per round
        traffic = json.load(open(pmc))["corrected_bytes_per_step"]["total"]

    if rank == 0:
        total_envs = n * world
        out = {
            "metric": "env-steps/sec at N=65536 RandomMiniEnv, 1/2/4/8 MI355X; % HBM roofline",
            "value": total_envs * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C3: RandomMiniEnv seed-0 geometry, %d envs/GPU, tricycle dynamic model + PlanEnv "
                                   "odometry noise (on-device Philox), shared 183x183 costmap, reset on done, steady-state episode phases" % n,
                       "envs_total": total_envs, "envs_per_gpu": n, "actions": "float32 U(action_space), pre-staged",
                       "sharding": ("env blocks per rank, done masks ring-buffered on the device and all-gathered (%s) every "
                                    "8 steps, overlapped with the next steps" % ("RCCL" if collective == "nccl" else collective))
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "step_fast_pair_kernel + step_pending_kernel (one step = these two launches)",
                         "kernel_ms": step_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * n,
                         "traffic_source": "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)",
                         "note": "shared-map config is latency-bound (SURVEY 8d; SQ counters in profiles/r01_step_alu_pmc.json: "
                                 "waves wait 73 % of their cycles at 2 waves per SIMD): the HBM fraction is low by "
                                 "construction; see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g)
        if world == 1 and not args.no_aux:
            try:  # informational only: never let it cost the metric line
                out["aux"] = aux_measurements(env, pool, n)
            except Exception as exc:  # noqa: BLE001
                out["aux"] = {"error": repr(exc)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
