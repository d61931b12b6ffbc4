#!/bin/bash
# where the pool step's bytes go (diagnostic build): FETCH_SIZE / WRITE_SIZE under the ablation flags
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/ablp_$c -o p -- python3 tools/pmc_ablate_run.py pool > gpurun_out/ablp_$c.log 2>&1
done
python3 tools/pmc_ablate_run.py --read gpurun_out/ablp_ | tee gpurun_out/pool_ablate_bytes.txt
