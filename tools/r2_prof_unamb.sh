mkdir -p gpurun_out/r2d; O=$PWD/gpurun_out/r2d; R=$PWD
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spaced or fw or unamb" > $O/parity.log 2>&1; tail -2 $O/parity.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unamb -- python3 $R/tools/other_rates.py > $O/other_rates.txt 2>&1
cat $O/other_rates.txt | grep -v amdgpu.ids
find $O/unamb -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
cd $R; python tools/sweep.py --mode spaced --k 21 --tiles 0,2048,4096 > $O/c5.txt 2>&1; grep -v amdgpu $O/c5.txt
