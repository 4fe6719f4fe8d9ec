timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-clips 0 --layers-json gpurun_out/c133_0.json > gpurun_out/c133_0.log 2>&1 || echo fail 0
for d in 1 2 4 8 3; do
  AF_HIP_LIB=$PWD/tools/tmp/libaf_c$d.so timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-clips 0 --layers-json gpurun_out/c133_$d.json > gpurun_out/c133_$d.log 2>&1 || echo fail $d
done
