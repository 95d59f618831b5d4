#!/bin/bash
# Side probes on the GPU box: cross-CU exchange cost, SQ counters of the matrix build.  Usage: bash scripts/probe_round.sh r02
R=${1:-r02}
OUT=$PWD/gpurun_out/${R}_probes
mkdir -p $OUT
REPO=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/xcu_sync_probe tests/probes/xcu_sync_probe.hip 2> $OUT/xcu_build.log \
&& timeout -k 10 120 $OUT/xcu_sync_probe > $OUT/xcu_sync.json 2> $OUT/xcu_sync.err && cat $OUT/xcu_sync.json \
&& cd /tmp && export TMPDIR=/tmp \
&& timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dm_stats -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_stats.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/dm_sq1 -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_sq1.log 2>&1 \
&& timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/dm_sq2 -- python3 $REPO/scripts/dm_build_once.py > $OUT/dm_sq2.log 2>&1 ; echo "rc=$?"; tail -3 $OUT/dm_sq2.log
