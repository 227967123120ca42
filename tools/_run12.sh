cd /root/repo
R=/root/repo/gpurun_out/batch
mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "dense" 2>&1 | tail -3
for d in 0 -1; do python3 tools/batch_once.py --dense $d --reps 6 2>&1 | grep -v amdgpu.ids; done
for p in 1 2 4; do python3 tools/batch_once.py --passes $p --reps 6 2>&1 | grep -v amdgpu.ids; done
KERNEL=ragged_kernel bash tools/pmc_once.sh $R/pmc_d0 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" python3 /root/repo/tools/batch_once.py --reps 2 > $R/pmc_d0.txt 2>&1
cat $R/pmc_d0.txt
