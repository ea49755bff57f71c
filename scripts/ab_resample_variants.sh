#!/bin/bash
# A/B of resample.hip build variants (scripts/build_variant.sh), bench.py --workload resample, interleaved processes
for i in 1 2 3; do
  for v in base "$@"; do
    if [ "$v" == base ]; then unset TSDGPU_LIB; else export TSDGPU_LIB=libtsd_amd/lib/variants/libtsdgpu_$v.so; fi
    python bench.py --workload resample --steps 40 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resample $v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
