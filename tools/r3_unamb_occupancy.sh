for rep in 1 2; do
for lib in "" ${LIBS:-u7 u8}; do
  for leg in u31 u21; do
    if [ -z "$lib" ]; then python3 tools/leg.py --leg $leg --alloc arena:0 2>/dev/null | grep -v "arena map" | sed "s/^/product /"
    else KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_$lib.so python3 tools/leg.py --leg $leg --alloc arena:0 2>/dev/null | grep -v "arena map" | sed "s/^/$lib      /"; fi
  done
done
done
