mkdir -p gpurun_out/r2c; O=gpurun_out/r2c
python tools/diag_c3.py > $O/diag_c3.txt 2>&1; cat $O/diag_c3.txt
D=kmers.jl_amd/csrc/libkmers_hip.so
python tools/sweep.py --mode spaced --k 21 --tiles 3072,4096,5120 --libs $D,tools/libkmers_b128.so > $O/c5.txt 2>&1; cat $O/c5.txt
python tools/sweep.py --mode spaced --k 21 --src-bits 2 --tiles 2048,4096,5120 --libs $D > $O/c5_2bit.txt 2>&1; cat $O/c5_2bit.txt
