#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash tools/profile.sh <label> [bench args...]
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<label>/ and a compact
# summary gpurun_out/prof_<label>/summary.txt (copy what should be judged into profiles/).
set -e
LABEL=${1:-run}; shift || true
ARGS=${@:---steps 30 --warmup 5 --no-cpu-baseline}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$LABEL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log || true
# (the counter passes serialise dispatches: the update step is launched in line there, ope_icp_params.update_launch; the search kernel is the same)
pmc() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $ARGS --update-launch in-line > /dev/null 2> $OUT/pmc_$name.log || echo "pmc $name failed" >> $OUT/errors.txt; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pmc sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_SCA
pmc tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TOTAL_READ TCP_PENDING_STALL_CYCLES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT TCC_MISS TCC_REQ
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
