"""Sharding the env batch over the GPUs of one node: one process per GPU, envs split by contiguous index blocks,
nothing crosses GPUs on the data path except the all-gather of the uint8 done mask (RCCL over xGMI when the backend is
"nccl"; "gloo" on CPU for tests) -- per step, or of a ring of the last K masks every K steps (bench.py).  The reference has no counterpart: its envs are independent objects.
"""
import os

# the host driver of this pool only supports dmabuf IPC (RCCL / CUDA-tensor sharing across processes); must be in
# the environment before the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def env_block(n_total, rank, world_size):
    """Contiguous block of env indices owned by `rank`: (first, count).  Remainder goes to the lowest ranks."""
    base, rem = divmod(int(n_total), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def init_from_env(backend=None, timeout_s=180, force=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run, or bench.py's own
    launcher).  Returns (rank, world_size, local_rank).

    A single rank needs no process group and gets none -- unless `force` (or BCP_DIST_FORCE=1) asks for one: the group,
    the communicator and every collective of DoneGather then run for real at world size 1, which is how the RCCL
    branch is exercised on a box with one GPU (tests/test_gpu_sharding.py).

    The transport is an explicit choice: `backend`, else BCP_DIST_BACKEND, else "nccl" (RCCL over xGMI) -- there is no
    fallback.  RCCL needs one GPU per rank; with fewer GPUs than ranks the call fails at once and names the rehearsal
    transport (BCP_DIST_BACKEND=gloo: the done masks then travel through the host).  A process group is initialised
    ONCE per process: a second init_process_group on the same TCPStore after a failed one finds the first attempt's
    rendezvous keys and the ranks wait on each other until the store times out (the hang of round 1's 2-rank rehearsal).
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("BCP_DIST_FORCE", "0") not in ("", "0")
    if (world > 1 or force) and not dist.is_initialized():
        import datetime
        if backend is None:
            backend = os.environ.get("BCP_DIST_BACKEND") or "nccl"
        if backend not in ("nccl", "gloo"):
            raise RuntimeError("BCP_DIST_BACKEND must be 'nccl' (RCCL) or 'gloo' (rehearsal), got %r" % (backend,))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29517")
        kwargs = {}
        if backend == "nccl":
            # The only collective of this job is a ~1 MB all-gather per 128 steps, and every compute unit its kernel holds is
            # one the step kernel's workgroups queue for: cap RCCL at RCCL_CHANNELS channels (= workgroups = CUs, of 256)
            # unless the job's environment says otherwise.  Must be set before the communicator exists.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(RCCL_CHANNELS))
            os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
            gpus = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            if gpus < local_world:
                raise RuntimeError(
                    "RCCL needs one GPU per rank: %d ranks on this node, %d GPU(s) visible.  To rehearse the sharded "
                    "path on fewer GPUs choose the host transport explicitly: BCP_DIST_BACKEND=gloo" % (local_world, gpus))
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    return rank, world, local_rank


RCCL_CHANNELS = 4   # workgroups RCCL's all-gather may run with (init_from_env)


def geometry_digest(*arrays):
    """SHA-256 over the bytes (and shapes / dtypes) of the arrays that make up a rank's replicated geometry -- shared
    costmap, origin, resolution, shared path --, as four int64 (a tensor that any backend can gather)."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(("%s%s|" % (a.dtype.str, a.shape)).encode())
        h.update(a.tobytes())
    return torch.from_numpy(np.frombuffer(h.digest(), dtype=np.int64).copy())


def check_same_geometry(digest, group=None, what="the shared costmap / path"):
    """Every rank steps replicas of ONE geometry (the shared costmap and path are replicated, each rank builds
    its copy from its own inputs).  All-gather the ranks' digests and raise on every rank if any differs from rank 0's:
    a job whose ranks disagree would train on a mixture nobody asked for, silently.  No-op without a process group."""
    if not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    mine = digest.to(torch.device("cuda", torch.cuda.current_device())) if on_gpu else digest.cpu()
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
    rows = out.view(world, -1).cpu()
    bad = [r for r in range(world) if not torch.equal(rows[r], rows[0])]
    if bad:
        raise RuntimeError("ranks %s hold a different copy of %s than rank 0: refusing to step a sharded batch on mixed geometry"
                           % (bad, what))


def broadcast_geometry(arrays, src=0, group=None):
    """Rank `src`'s arrays for everybody (ncclBroadcast of SURVEY 8e; gloo on the CPU): a list of numpy arrays whose
    shapes and dtypes all ranks already agree on (they come from the same configuration).  Returns numpy arrays."""
    import numpy as np
    if not dist.is_initialized():
        return [np.ascontiguousarray(a) for a in arrays]
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    out = []
    for a in arrays:
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev)
        dist.broadcast(t, src=src, group=group)
        out.append(t.cpu().numpy().view(a.dtype).reshape(a.shape))
    return out


def local_device(local_rank):
    """GPU of this rank: its own with RCCL; rank % GPUs in a gloo rehearsal (several ranks may share one GPU)."""
    return local_rank % max(1, torch.cuda.device_count())


def _native(t):
    """libbcplan for device tensors (bcp_pack_mask_bits / bcp_unpack_mask_bits: one pass); tensor arithmetic on the CPU"""
    if not t.is_cuda:
        return None
    from . import _lib
    return _lib.load()


def pack_mask_bits(mask, out=None):
    """uint8 mask [8 k] (non-zero = set) -> uint8 [k]: bit b of byte j = mask[8 j + b] (include/bcplan.h: bcp_pack_mask_bits)."""
    lib = _native(mask)
    if lib is not None and mask.numel() % 32 == 0 and mask.is_contiguous():
        from . import _lib
        if out is None:
            out = torch.empty(mask.numel() // 8, dtype=torch.uint8, device=mask.device)
        _lib.check(lib.bcp_pack_mask_bits(mask.data_ptr(), mask.numel(), out.data_ptr(),
                                          torch.cuda.current_stream(mask.device).cuda_stream))
        return out
    w = (1 << torch.arange(8, device=mask.device, dtype=torch.int32)).to(torch.uint8)
    bits = ((mask.view(-1, 8) != 0).to(torch.uint8) * w).sum(dim=1, dtype=torch.uint8)
    return bits if out is None else out.copy_(bits)


def unpack_mask_bits(bits):
    """inverse of pack_mask_bits: uint8 [k] -> uint8 mask [8 k] of zeros and ones"""
    lib = _native(bits)
    if lib is not None and bits.numel() % 4 == 0 and bits.is_contiguous():
        from . import _lib
        out = torch.empty(bits.numel() * 8, dtype=torch.uint8, device=bits.device)
        _lib.check(lib.bcp_unpack_mask_bits(bits.data_ptr(), out.numel(), out.data_ptr(),
                                            torch.cuda.current_stream(bits.device).cuda_stream))
        return out
    sh = torch.arange(8, device=bits.device, dtype=torch.uint8)
    return ((bits.view(-1, 1) >> sh) & 1).reshape(-1)


class DoneGather(object):
    """All-gather of the per-rank done mask into the global mask [world_size * n_local] (uint8).

    Equal shard sizes are required (all_gather_into_tensor); pad the last shard if the batch does not divide.
    `packed` (pipelined form only): the mask crosses the links as one BIT per env -- pack_mask_bits before the collective,
    result() unpacks on request -- an eighth of the bytes for an RCCL kernel that shares the compute units with the steps
    (n_local must be a multiple of 8)."""

    def __init__(self, n_local, device, group=None, packed=False):
        self.group = group
        # with a process group every gather is a collective, also at world size 1 (init_from_env(force=True));
        # without one there is nothing to gather from
        self.collective = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.collective else 1
        self.n_local = int(n_local)
        self.packed = bool(packed)
        if self.packed and self.n_local % 8:
            raise ValueError("DoneGather(packed=True) needs a multiple of 8 envs per rank, got %d" % self.n_local)
        self.out = torch.zeros(self.world * (self.n_local // 8 if self.packed else self.n_local), dtype=torch.uint8, device=device)

    def __call__(self, done_local, async_op=False):
        if self.packed:
            raise ValueError("DoneGather(packed=True) is the pipelined form: launch() / result()")
        if not self.collective:
            self.out.copy_(done_local)
            return None if async_op else self.out
        if self.out.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal path (several ranks sharing one GPU): gloo gathers on the host
            host = torch.empty(self.out.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, done_local.cpu(), group=self.group)
            self.out.copy_(host)
            return None if async_op else self.out
        work = dist.all_gather_into_tensor(self.out, done_local, group=self.group, async_op=async_op)
        return work if async_op else self.out

    # ---- pipelined form: the gather of step t overlaps the kernels of step t+1 -------------------------------
    def launch(self, done_local):
        """Start gathering this step's done mask without stalling the compute stream.  The mask is first copied to
        one of two staging buffers (the step kernel overwrites `done_local` next step), then all-gathered
        asynchronously; the result of THIS call is what `result()` returns after the next `launch()` / `flush()`."""
        if not hasattr(self, "_stage"):
            self._stage = [torch.empty(self.n_local // 8 if self.packed else self.n_local, dtype=torch.uint8, device=done_local.device)
                           for _ in range(2)]
            self._outs = [self.out, torch.empty_like(self.out)]
            self._work = [None, None]
            self._k = 0
        k = self._k
        if self._work[k] is not None:  # buffer k was used two launches ago: its gather must have completed
            self._work[k].wait()
            self._work[k] = None
        if self.packed:
            pack_mask_bits(done_local, out=self._stage[k])
        else:
            self._stage[k].copy_(done_local)
        if not self.collective:
            self._outs[k].copy_(self._stage[k])
        elif self.out.is_cuda and dist.get_backend(self.group) == "gloo":
            host = torch.empty(self.out.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, self._stage[k].cpu(), group=self.group)
            self._outs[k].copy_(host)
        else:
            self._work[k] = dist.all_gather_into_tensor(self._outs[k], self._stage[k], group=self.group, async_op=True)
        self._last = k
        self._k = 1 - k
        return k

    def result(self, unpack=True):
        """Global done mask of the most recent launch() (waits for its gather).  packed: a uint8 mask of zeros and ones
        [world_size * n_local], or with unpack=False the gathered bits as they arrived (pack_mask_bits per rank)."""
        k = self._last
        if self._work[k] is not None:
            self._work[k].wait()
            self._work[k] = None
        return unpack_mask_bits(self._outs[k]) if (self.packed and unpack) else self._outs[k]

    def flush(self):
        for k in range(2):
            if getattr(self, "_work", [None, None])[k] is not None:
                self._work[k].wait()
                self._work[k] = None
