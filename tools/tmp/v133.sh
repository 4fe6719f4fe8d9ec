for v in v1 v2; do
  AF_HIP_LIB=$PWD/tools/tmp/libaf_$v.so timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-clips 0 --layers-json gpurun_out/c133_$v.json > gpurun_out/c133_$v.log 2>&1 || echo fail $v
done
