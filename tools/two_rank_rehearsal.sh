#!/bin/bash
# Two ranks of bench.py on ONE GPU: the rehearsal of the sharded path this box allows (RCCL needs a GPU per rank, so
# the transport is chosen explicitly: gloo, the done masks travel through the host).  bench.py starts its ranks itself.
# Round 1's rehearsal hung: a launcher (not kept) caught the failed RCCL set-up and called init_process_group a second
# time ("falling back to gloo") on the same TCPStore; the second rendezvous met the first attempt's keys and the ranks
# waited on each other until gpurun's limit.  There is no such fallback any more (bc_gym_planning_env_amd/distributed.py).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
BCP_DIST_BACKEND=gloo timeout -k 10 420 python bench.py --gpus 2 --steps 40 --warmup 8 --no-aux --no-cpu-baseline \
    > gpurun_out/two_rank_rehearsal.json 2> gpurun_out/two_rank_rehearsal.err
rc=$?
echo "rehearsal rc=$rc"
cat gpurun_out/two_rank_rehearsal.json
# without the explicit transport the same command must fail fast and say why
start=$(date +%s)
timeout -k 10 120 python bench.py --gpus 2 --steps 4 --warmup 1 --no-aux --no-cpu-baseline \
    > gpurun_out/two_rank_refused.json 2> gpurun_out/two_rank_refused.err
rc2=$?
echo "without BCP_DIST_BACKEND: rc=$rc2 after $(( $(date +%s) - start )) s"
grep -m1 "RCCL needs one GPU per rank" gpurun_out/two_rank_refused.err || true
exit $rc
