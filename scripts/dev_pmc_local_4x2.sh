# PMC of the panel products at the 4x2 grid's local shape of config 4 (H_loc 16384 x 32768 complex, 256-column panels, K pieces
# for shared-chip launches min_rounds = 4), op C (column -> row step) and op N (row -> column step)
G="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"
for op in C N; do
  for r in 4 0; do
    PMC_GROUPS="$G" PMC_DRIVER=scripts/dev_panel_only.py bash scripts/prof_pmc.sh gpurun_out/r03_pmc_local_4x2_op${op}_rounds${r}.txt z $op 16384 32768 256 $r 10
  done
done
