set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3p}
mkdir -p $O/tc $O/pf_fetch $O/pf_write $O/kt $O/fetch $O/write
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tc -- python3 $R/tools/time_configs.py val,dense,sweep > $O/tc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_fetch -- python3 $R/tools/pmc_val_nms.py > $O/pf_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pf_write -- python3 $R/tools/pmc_val_nms.py > $O/pf_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --pre-roll 0 --post-roll 0 > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --pre-roll 0 --post-roll 0 > $O/write.log 2>&1
cd $R
timeout -k 10 300 python3 bench.py > $O/bench_line.json 2> $O/bench.err
timeout -k 10 200 python3 tools/time_configs.py val,dense,sweep --iters 20 > $O/time_configs.txt 2>&1
tail -1 $O/bench_line.json
cat $O/time_configs.txt
