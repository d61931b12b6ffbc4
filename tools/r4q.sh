#!/bin/bash
# round-4 profiles: run from the repo root on the GPU box; everything lands in gpurun_out/ (copy to profiles/r04_*)
bash tools/step_pmc.sh > gpurun_out/r4q_step_pmc.log 2>&1; echo "step_pmc rc=$?"
bash tools/c4_pmc.sh > gpurun_out/r4q_c4_pmc.log 2>&1; echo "c4_pmc rc=$?"
bash tools/pool_pmc.sh > gpurun_out/r4q_pool_pmc.log 2>&1; echo "pool_pmc rc=$?"
bash tools/step_alu_pmc.sh > gpurun_out/r4q_alu.log 2>&1; echo "alu rc=$?"
bash tools/step_alu_pmc.sh c4 > gpurun_out/r4q_alu_c4.log 2>&1; echo "alu c4 rc=$?"
bash tools/ego_aisle_pmc.sh > gpurun_out/r4q_ego_aisle.log 2>&1; echo "ego aisle rc=$?"
bash tools/ego_pmc.sh > gpurun_out/r4q_ego.log 2>&1; echo "ego rc=$?"
