#!/bin/bash
# Runs tools/micro/slot_forward (built beforehand) plainly and under a kernel trace; results under gpurun_out/r3_slot/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_slot; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 $R/tools/micro/slot_forward 300 > $O/slot_forward.txt 2>&1 || { cat $O/slot_forward.txt; exit 1; }
cat $O/slot_forward.txt
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $R/tools/micro/slot_forward 100 > $O/trace_log.txt 2>&1 || { tail -5 $O/trace_log.txt; exit 1; }
python3 $R/tools/kernel_trace_by_grid.py $O/trace > $O/kernels.txt 2>&1; cat $O/kernels.txt
