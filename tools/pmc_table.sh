#!/bin/bash
# Counter passes over ONE python script and a derived table per kernel (GPU box):
#   bash tools/pmc_table.sh <kernel-name-substring> <script.py> [script args...]     (environment variables pass through)
# Five rocprofv3 --pmc passes (8 SQ slots + GRBM per pass); the program itself directly behind `--`.
pat=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
script=$1; shift
case "$script" in /*) ;; *) script="$root/$script" ;; esac
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmct*
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmct$i -o p -- python3 "$script" "$@" > /dev/null 2>&1
done
python3 "$root/tools/pmc_table.py" "$pat" /tmp/pmct*
