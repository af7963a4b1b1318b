mkdir -p gpurun_out/r05c; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cylinder or cyl" > gpurun_out/r05c/cyltests.log 2>&1; tail -2 gpurun_out/r05c/cyltests.log
export CAPE_HIP_LIB=$PWD/rgb-d-slam_amd/lib/exp/libcape_prof.so
CAPE_RESUME=group CAPE_PHASES_GROUP=1 python profiles/grow_phases.py 4096 room cyl > gpurun_out/r05c/phases_group.txt 2>&1
CAPE_PHASES_GROUP=1 python profiles/grow_phases.py 1024 tunnel cyl 1280 960 > gpurun_out/r05c/phases_1280.txt 2>&1
unset CAPE_HIP_LIB
for m in wave group; do CAPE_RESUME=$m python bench.py --no-cpu-baseline --steps 10 --warmup 3 --cylinders --no-polygons --no-parity-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['roofline']['kernel_ms'])"; done
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --cylinders --no-polygons --no-parity-check --width 1280 --height 960 --frames 1024 --scene tunnel 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1280', d['value'], d['roofline']['kernel_ms'])"
