# Dev tool: HBM traffic counters of the level-1 conv (one counter group per pass, as MI355X_MICROARCH.md prescribes)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_conv_d/g$i -- python3 $R/tools/prof_conv_l1.py 3 > /dev/null 2>&1
  echo "group $i ($grp) rc=$?"
done
