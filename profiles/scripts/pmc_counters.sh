#!/bin/bash
# SQ / TA / TCP counters of every launch of a config's forward (one rocprofv3 --pmc pass per counter group; kernel-trace only).
# usage: bash profiles/scripts/pmc_counters.sh <cfg> -> gpurun_out/ctr_<cfg>/<group>/...counter_collection.csv.gz
cfg=${1:-en_s0}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
o=gpurun_out/ctr_$cfg
mkdir -p $o
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $o/g$i -- python profiles/scripts/pmc_forward.py --config $cfg --forwards 3 --ops-json $o/ops.json > $o/g$i.log 2>&1
  tail -1 $o/g$i.log
  find $o/g$i -name "*counter_collection.csv" -exec gzip -9 {} \;
  find $o/g$i -type f ! -name "*counter_collection.csv.gz" -delete
done
ls -R $o | head -30
