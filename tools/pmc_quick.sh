#!/bin/bash
# quick SQ counter pass: bash tools/pmc_quick.sh <label>
LABEL=${1:-q}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmcq_$LABEL; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq1 -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-coarse > /dev/null 2> $OUT/log.txt
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-coarse > /dev/null 2>> $OUT/log.txt
python3 $REPO/tools/summarize_prof.py $OUT
