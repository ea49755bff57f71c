#!/bin/bash
# usage: scripts/pmc_collect.sh <tag> <script.py> <args...>      (script relative to the repo root)
# One rocprofv3 --pmc pass per counter group (counters are never combined with tracing of other
# domains), outputs under gpurun_out/pmc_<tag>_<group>/ ; summarise with scripts/pmc_summary.py
set -e
TAG=$1; shift
SCRIPT=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
if [ -n "$PMC_TRAFFIC_ONLY" ]; then
  GROUPS_=("FETCH_SIZE" "WRITE_SIZE")
else
  GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE")
fi
i=0
for g in "${GROUPS_[@]}"; do
  out=$ROOT/gpurun_out/pmc_${TAG}_$i
  rm -rf $out
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python3 $ROOT/$SCRIPT "$@" > $out.log 2>&1) || echo "group $i failed"
  i=$((i+1))
done
python3 $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/pmc_${TAG}_* | tee $ROOT/gpurun_out/pmc_${TAG}_summary.txt
