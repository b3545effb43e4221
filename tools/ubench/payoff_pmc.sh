cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/paypmc; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p -- python $R/tools/ubench/payoff_probe.py > $O/$tag.log 2>&1; echo "$tag rc=$?"
done
ls $O
