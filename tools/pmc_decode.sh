set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3dec
mkdir -p $O/sqa $O/sqb
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sqa -- python3 $R/tools/pmc_decode.py > $O/sqa.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sqb -- python3 $R/tools/pmc_decode.py > $O/sqb.log 2>&1
cd $R
python3 tools/pmc_by_grid.py $O/decode.json $O/sqa $O/sqb --only k_yolo_decode
