cd /root/repo
mkdir -p gpurun_out/batch
export TMPDIR=/tmp
python3 tools/batch_once.py --reps 5 2>&1 | grep -v amdgpu.ids > gpurun_out/batch/plain.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/batch/trace -- python3 /root/repo/tools/batch_once.py > /root/repo/gpurun_out/batch/trace_run.txt 2>&1 )
cp $(find gpurun_out/batch/trace -name "*kernel_stats.csv" | head -1) gpurun_out/batch/kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/batch/trace
KERNEL=kernel bash tools/pmc_once.sh gpurun_out/batch/pmc_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" python3 /root/repo/tools/batch_once.py --reps 2 > gpurun_out/batch/pmc_sq.txt 2>&1
KERNEL=kernel bash tools/pmc_once.sh gpurun_out/batch/pmc_f "FETCH_SIZE" python3 /root/repo/tools/batch_once.py --reps 2 > gpurun_out/batch/pmc_fetch.txt 2>&1
KERNEL=kernel bash tools/pmc_once.sh gpurun_out/batch/pmc_w "WRITE_SIZE" python3 /root/repo/tools/batch_once.py --reps 2 > gpurun_out/batch/pmc_write.txt 2>&1
KERNEL=kernel bash tools/pmc_once.sh gpurun_out/batch/pmc_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python3 /root/repo/tools/batch_once.py --reps 2 > gpurun_out/batch/pmc_sq2.txt 2>&1
for p in 1 2 3 4 6 8; do python3 tools/batch_once.py --passes $p 2>&1 | grep -v amdgpu.ids; done > gpurun_out/batch/passes.txt
cat gpurun_out/batch/plain.txt gpurun_out/batch/kernel_stats.csv gpurun_out/batch/pmc_sq.txt gpurun_out/batch/pmc_fetch.txt gpurun_out/batch/pmc_write.txt gpurun_out/batch/pmc_sq2.txt gpurun_out/batch/passes.txt
