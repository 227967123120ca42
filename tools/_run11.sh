cd /root/repo
R=/root/repo/gpurun_out/batch
mkdir -p $R
export TMPDIR=/tmp
for d in 0 -1; do python3 tools/batch_once.py --dense $d --reps 6 2>&1 | grep -v amdgpu.ids; done
for d in 0 -1; do
KERNEL=ragged_kernel bash tools/pmc_once.sh $R/pmc_d$d "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" python3 /root/repo/tools/batch_once.py --reps 2 --dense $d > $R/pmc_d$d.txt 2>&1
cat $R/pmc_d$d.txt
done
