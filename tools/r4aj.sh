#!/bin/bash
# profiles of the round's last build, part 2: SQ counters of step_local_kernel on C3 and C4, N sweep, soak against the oracle
O=gpurun_out/r4aj; mkdir -p $O; rm -f $O/*
bash tools/step_alu_pmc.sh > $O/alu_c3.log 2>&1; echo "alu c3 rc=$?" >> $O/rc.txt
bash tools/step_alu_pmc.sh c4 > $O/alu_c4.log 2>&1; echo "alu c4 rc=$?" >> $O/rc.txt
python tools/n_sweep.py > $O/n_sweep.txt 2>&1; echo "n_sweep rc=$?" >> $O/rc.txt
python tests/soak.py 1500 32768 13 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/rc.txt
python tools/soak_endless.py >> $O/soak.txt 2>&1; echo "soak endless rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 12 $O/soak.txt; cat $O/n_sweep.txt | tail -n 12
