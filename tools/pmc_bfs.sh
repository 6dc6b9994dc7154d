R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_bfs/g$i -- python3 $R/tools/prof_bfs.py 2 > /dev/null 2>&1
  echo "group $i rc=$?"
done
