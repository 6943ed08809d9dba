# Kernel trace + four counter passes over tools/pmc_kernels.py (run on the GPU box from the repository root):
#   bash tools/collect_kernel_counters.sh [sections] [tag]
# then, here:  python tools/kernel_trace_by_grid.py gpurun_out/<tag>/kt profiles/<tag>_kernels_by_size.csv
#              python tools/pmc_by_grid.py profiles/<tag>_pmc_kernels.json gpurun_out/<tag>/fetch gpurun_out/<tag>/write gpurun_out/<tag>/sqa gpurun_out/<tag>/sqb
set -e
R=$GRAFT_REPO_ROOT
S=${1:-pairwise,iou,cnloss,cndecode,convval,head}
O=$R/gpurun_out/${2:-r3k}
mkdir -p $O/kt $O/fetch $O/write $O/sqa $O/sqb
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/pmc_kernels.py $S 6 > $O/kt.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/pmc_kernels.py $S 2 > $O/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/pmc_kernels.py $S 2 > $O/write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sqa -- python3 $R/tools/pmc_kernels.py $S 2 > $O/sqa.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sqb -- python3 $R/tools/pmc_kernels.py $S 2 > $O/sqb.log 2>&1
cd $R
python3 tools/kernel_trace_by_grid.py $O/kt > $O/by_size.txt
tail -60 $O/by_size.txt
