#!/bin/bash
# SQ instruction counters of the accumulate kernel for an arbitrary probe: bash tools/pmc_script.sh <label> <script args...>
LABEL=$1; shift; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmcs_$LABEL; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $REPO/"$@" > /dev/null 2> $OUT/log.txt
python3 $REPO/tools/summarize_prof.py $OUT | grep -v "^==" 
