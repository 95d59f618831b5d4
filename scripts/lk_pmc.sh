#!/bin/bash
# SQ / L2 counters of the LK kernels at n = 13 509 (5 epochs).  Usage: bash scripts/lk_pmc.sh r02
R=${1:-r02}
OUT=$PWD/gpurun_out/${R}_lkpmc
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 $REPO/scripts/lk_profile.py > $OUT/sq.log 2>&1 \
&& timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 $REPO/scripts/lk_profile.py > $OUT/tcc.log 2>&1 \
&& timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tcp -- python3 $REPO/scripts/lk_profile.py > $OUT/tcp.log 2>&1; echo rc=$?; tail -2 $OUT/tcp.log
