for i in 1 2 3; do
  for v in 0 16 32 8; do
    TSDGPU_RS_DYN=$v python bench.py --workload resample --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rs dyn=$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
