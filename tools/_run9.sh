cd /root/repo
R=/root/repo/gpurun_out/batch
mkdir -p $R
export TMPDIR=/tmp
KERNEL=ragged bash tools/pmc_once.sh $R/pmc_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" python3 /root/repo/tools/batch_once.py --reps 2 > $R/pmc_sq.txt 2>&1
KERNEL=ragged bash tools/pmc_once.sh $R/pmc_f "FETCH_SIZE" python3 /root/repo/tools/batch_once.py --reps 2 > $R/pmc_fetch.txt 2>&1
KERNEL=ragged bash tools/pmc_once.sh $R/pmc_w "WRITE_SIZE" python3 /root/repo/tools/batch_once.py --reps 2 > $R/pmc_write.txt 2>&1
KERNEL=ragged bash tools/pmc_once.sh $R/pmc_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_WAIT_ANY" python3 /root/repo/tools/batch_once.py --reps 2 > $R/pmc_sq2.txt 2>&1
KERNEL=ragged bash tools/pmc_once.sh $R/pmc_sq3 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY" python3 /root/repo/tools/batch_once.py --reps 2 > $R/pmc_sq3.txt 2>&1
cat $R/pmc_sq.txt $R/pmc_fetch.txt $R/pmc_write.txt $R/pmc_sq2.txt $R/pmc_sq3.txt | grep -v "^$"
