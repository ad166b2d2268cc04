O=$PWD/gpurun_out/dbg; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/_p -o pmc -- python $REPO/tools/time_gene_kernel.py > $O/pmc_gene.log 2>&1)
  f=$(find $O/_p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/summarize_pmc.py "$f" | grep -A8 "k_gene_fused"
  rm -rf $O/_p
done
