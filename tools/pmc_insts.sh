mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail > $R/gpurun_out/pmc/avail.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/gpurun_out/pmc/p1 --output-format csv -- python3 $R/tools/clip_debug.py 30 > $R/gpurun_out/pmc/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc/p2 --output-format csv -- python3 $R/tools/clip_debug.py 30 > $R/gpurun_out/pmc/p2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc/p3 --output-format csv -- python3 $R/tools/clip_debug.py 30 > $R/gpurun_out/pmc/p3.log 2>&1
tail -3 $R/gpurun_out/pmc/p*.log; find $R/gpurun_out/pmc -name "*.csv" | head
