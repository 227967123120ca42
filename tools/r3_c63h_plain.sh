for rep in 1 2; do for alloc in plain carve:40; do for shape in 0:0 128:768 256:1024 256:768 128:512; do
  python3 tools/leg.py --leg c63h --alloc $alloc --threads ${shape%%:*} --tile ${shape##*:} 2>/dev/null | grep -v "arena map"
done; done; done
