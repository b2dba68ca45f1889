#!/bin/bash
# Lab: PMC passes on the register-resident 128-channel conv (K2q) and on the patch kernel it replaces, one counter group
# per pass (kernel-trace only).  Usage (on the GPU box): tools/lab/k2q_pmc.sh ; output under gpurun_out/k2q_pmc/
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/k2q_pmc
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | cut -d' ' -f1)
  for r in 1 0; do
    JSPSR_CONV_RESIDENT128=$r rocprofv3 --pmc $grp -f csv -d $out/${tag}_r$r -o p -- python3 $root/tools/bench_conv.py fwd 1 bf16 x > $out/${tag}_r$r.log 2>&1 || echo "pass $tag r$r failed"
    python3 $root/tools/pmc_summary.py $out/${tag}_r$r conv >> $out/summary_r$r.txt
    find $out/${tag}_r$r -name "*.db" -delete
  done
done
cat $out/summary_r1.txt; echo ----; cat $out/summary_r0.txt
