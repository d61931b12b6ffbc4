"""One rank of a sharded batch (started by tests/test_gpu_sharding.py through tests/launcher.py, RANK / WORLD_SIZE /
MASTER_* in the environment): steps its block of envs on the GPU and all-gathers the done masks, per step and through
the ring of bench.py.  Rank 0 writes what it gathered to argv[1] (npz)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def global_actions(n_total, steps, seed=77):
    rng = np.random.RandomState(seed)
    lo, hi = np.array([np.pi / 30, -np.pi / 2]), np.array([np.pi / 6, np.pi / 2])
    a = rng.uniform(lo, hi, (steps, n_total, 2)).astype(np.float32)
    a[:, :, 0] = np.float32(0.5)   # fast: collisions and finished episodes within a few dozen steps
    return a


def make_shard(n_local, first, device, timeout=30):
    from util import GOLDEN
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(GOLDEN, "g8_traj_mini_03.npz"))
    res = float(g["resolution"])
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False,
                       iteration_timeout=timeout)
    return BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n_local, device=device,
                          auto_reset=True, env_id_base=first, seed=2024)


def main():
    out_path, n_total, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    from bc_gym_planning_env_amd import distributed as bdist
    rank, world, local_rank = bdist.init_from_env()
    device = bdist.local_device(local_rank)
    torch.cuda.set_device(device)
    first, count = bdist.env_block(n_total, rank, world)
    assert count * world == n_total
    env = make_shard(count, first, device)
    acts = global_actions(n_total, steps)
    dev = torch.device("cuda", device)
    every = 4
    ring = torch.zeros(every, count, dtype=torch.uint8, device=dev)
    per_step = bdist.DoneGather(count, dev)
    ring_gather = bdist.DoneGather(every * count, dev, packed=True)   # (as bench.py: one bit per env on the wire)
    got_step, got_ring = [], []
    for k in range(steps):
        a = torch.from_numpy(acts[k, first:first + count]).to(dev)
        env.step(a, done_out=ring[k % every])
        got_step.append(per_step(ring[k % every]).cpu().numpy().copy())
        if k % every == every - 1:
            ring_gather.launch(ring.view(-1))
            # [world][every][count] -> [every][world * count]
            r = ring_gather.result().cpu().numpy().reshape(world, every, count)
            got_ring.extend(np.concatenate([r[w, j] for w in range(world)]) for j in range(every))
    ring_gather.flush()
    env.check_errors()
    state = env.state.robot.cpu().numpy()
    # every rank contributes its final shard state (gathered on the host: test plumbing, not the data path)
    states = [None] * world
    dist.all_gather_object(states, state)
    if rank == 0:
        np.savez(out_path, done_per_step=np.stack(got_step), done_ring=np.stack(got_ring),
                 robot=np.concatenate(states, axis=1), backend=dist.get_backend())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
