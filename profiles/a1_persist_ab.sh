for rep in 1 2; do
for v in default a1p4 a1p5; do
  if [ $v = default ]; then unset CAPE_HIP_LIB; else export CAPE_HIP_LIB=$PWD/rgb-d-slam_amd/lib/exp/libcape_$v.so; fi
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-polygons --no-cylinders-on --parity-frames 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), d['roofline']['kernel_ms'], d['parity_check']['labels_equal'], d['parity_check']['segments_bitwise'])"
done; done
