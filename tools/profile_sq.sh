#!/bin/bash
# Where do the waves of the KDyn kernels spend their cycles?  SQ counters in separate --pmc passes of a short run -> gpurun_out/prof_sq_<N>/summary.txt
#   WAVE_CYCLES ~ WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY   (MI355X_MICROARCH.md)
# usage: tools/profile_sq.sh [npts] [iters]
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-128}; IT=${2:-4}
OUT=$R/gpurun_out/prof_sq_$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/prof_kdyn.py $N $IT > $OUT/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/p$i.log; }
done
cd $R
python3 tools/summarize_pmc.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 > $OUT/summary.txt
grep -A26 "kd_x_pass<[0-9]*, [23]" $OUT/summary.txt | head -120
