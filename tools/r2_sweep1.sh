mkdir -p gpurun_out/r2b; O=gpurun_out/r2b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity_default.log 2>&1; tail -3 $O/parity_default.log
KMERS_HIP_LIB=$PWD/tools/libkmers_wp.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/parity_wp.log 2>&1; tail -3 $O/parity_wp.log
./tools/hbm_ceiling > $O/ceiling.txt 2>&1; tail -20 $O/ceiling.txt
D=kmers.jl_amd/csrc/libkmers_hip.so
LIBS=$D,tools/libkmers_wp.so,tools/libkmers_b128.so,tools/libkmers_b512.so
python tools/sweep.py --bases 1250000000 --src-bits 2 --no-hash --tiles 1024,1536,2048,3072,4096 --libs $LIBS > $O/c3.txt 2>&1; cat $O/c3.txt
python tools/sweep.py --mode spaced --k 21 --tiles 1024,2048,3072,4096 --libs $LIBS > $O/c5.txt 2>&1; cat $O/c5.txt
python tools/sweep.py --tiles 512,1024,2048 --libs $LIBS > $O/c2.txt 2>&1; cat $O/c2.txt
