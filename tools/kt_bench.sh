set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O/kt
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1
cd $R
python3 tools/kernel_trace_by_grid.py $O/kt > $O/by_size.txt
cat $O/by_size.txt
tail -1 $O/kt.log | cut -c1-600
