O=$PWD/gpurun_out/r3c2; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2; do
  python3 tools/leg.py --leg c2 --alloc arena:0 2>> $O/err.txt | grep -v "arena map" | sed "s/^/arena default            /" >> $T
  for sb in 1 2; do
   for shape in "128 1536" "128 2048" "256 3072" "128 1024" "256 2048"; do
    set -- $shape
    python3 tools/leg.py --leg c2 --alloc arenacarve:0 --straddle --straddle-b $sb --split --threads $1 --tile $2 2>> $O/err.txt | grep -v "arena map\|straddle:" | sed "s/^/both across, b=$sb split /" >> $T
   done
  done
  for leg in c4; do
   python3 tools/leg.py --leg c4 --alloc arena:0 2>> $O/err.txt | grep -v "arena map" | sed "s/^/arena default            /" >> $T
   for shape in "128 512" "128 1024" "256 1024" "256 1536"; do
    set -- $shape
    python3 tools/leg.py --leg c4 --alloc arenacarve:0 --straddle --straddle-b 1 --split --threads $1 --tile $2 2>> $O/err.txt | grep -v "arena map\|straddle:" | sed "s/^/both across, b=1 split /" >> $T
   done
  done
done
cat $T; tail -3 $O/err.txt
