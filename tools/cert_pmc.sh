#!/bin/bash
# counters of the certifying instantiation over a long settled run (most of its launches answer everything from certificates):
#   bash tools/cert_pmc.sh      (GPU box; writes gpurun_out/pmc_cert.txt)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_cert
rm -rf $OUT; mkdir -p $OUT
pmc() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $REPO/tools/cert_probe.py auto 600 > $OUT/log_$name.txt 2>&1; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pmc sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc grbm GRBM_GUI_ACTIVE
python3 $REPO/tools/summarize_prof.py $OUT 2>&1 | grep -v "^==" > $REPO/gpurun_out/pmc_cert.txt
grep "kernel us" $OUT/log_sq1.txt >> $REPO/gpurun_out/pmc_cert.txt
rm -rf $OUT
