mkdir -p gpurun_out/r02l
for lib in osu_dreamer_amd/libosudreamer_hip.so gpurun_variants/libod_qkhead.so gpurun_variants/libod_qkbpw2.so gpurun_variants/libod_qkbpw4.so gpurun_variants/libod_qkbpw16.so gpurun_variants/libod_qkbpw32.so; do
  echo "== $lib"; OSU_DREAMER_HIP_LIB=$PWD/$lib python tools/microbench.py rope --iters 20
done
echo "== L=32768 B=1 in-tree"; python tools/microbench.py rope --iters 20 --B 1 --L 32768
echo "== L=32768 B=1 head"; OSU_DREAMER_HIP_LIB=$PWD/gpurun_variants/libod_qkhead.so python tools/microbench.py rope --iters 20 --B 1 --L 32768
