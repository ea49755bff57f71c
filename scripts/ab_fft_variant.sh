for i in 1 2; do
  for w in base f1mp; do
    if [ "$w" == base ]; then unset TSDGPU_LIB; else export TSDGPU_LIB=libtsd_amd/lib/variants/libtsdgpu_$w.so; fi
    echo "$w"; python scripts/perf_fft_sizes.py 2097152 4194304 2>/dev/null
    python bench.py --workload fft --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
