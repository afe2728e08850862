#!/bin/bash
# round 6: does `rank_emulation.py 2 4 8` in ONE process return? (round 5: once it did not, for 30 minutes). Each attempt is a fresh child under a hard limit;
# the tool prints a time-stamped phase line per step and, shortly before its own alarm, every thread's Python stack.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do
  echo "=== attempt $i ==="
  t0=$(date +%s)
  EMU_TIMEOUT=${EMU_TIMEOUT:-240} timeout -s KILL 300 python tools/rank_emulation.py 2 4 8 > gpurun_out/r06_emu_repro_$i.json 2> gpurun_out/r06_emu_repro_$i.log
  echo "rc $? in $(( $(date +%s) - t0 )) s"
  tail -4 gpurun_out/r06_emu_repro_$i.log
done
