#!/bin/bash
# A/B of a variant build of the library against the default one, interleaved processes on one box:
#   scripts/ab_variant.sh NAME [bench.py args]     (libtsd_amd/lib/variants/libtsdgpu_NAME.so, see scripts/build_variant.sh)
v=$1; shift
for i in 1 2 3; do
  for w in base $v; do
    if [ "$w" == base ]; then unset TSDGPU_LIB; else export TSDGPU_LIB=libtsd_amd/lib/variants/libtsdgpu_$w.so; fi
    python bench.py --steps 200 --warmup 50 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
