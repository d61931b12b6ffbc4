"""Step time of the C3 geometry at small batch sizes: launched step by step (libbcplan's own timing loop: no Python
between the launches), from Python (env.step per step), and replayed from a captured HIP graph of 32 steps."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

for n in (256, 1024, 4096, 16384, 32768, 65536):
    env, g = bench.make_env(n, 0, 0, 1)
    rng = np.random.RandomState(0)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
    for k in range(1200):
        env.step(pool[k % 8])
    torch.cuda.synchronize()
    ms = min(env.time_steps(pool[i % 8], 50) for i in range(4))
    # from Python, one env.step() per step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(512):
        env.step(pool[k % 8])
    torch.cuda.synchronize()
    ms_py = (time.perf_counter() - t0) / 512 * 1e3
    # a captured graph of 32 steps
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for k in range(32):
                env.step(pool[k % 8])
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(16):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms_graph = e0.elapsed_time(e1) / (16 * 32)
    print("%6d envs: launched %.4f ms/step, python loop %.4f, graph replay %.4f  (%.3e env-steps/s)"
          % (n, ms, ms_py, ms_graph, n / min(ms, ms_graph) * 1e3), flush=True)
