#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <round-tag>
# Produces under gpurun_out/: per-workload bench JSON lines, rocprofv3 --kernel-trace --stats
# summaries of the same bench command, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs).
# Copy what is to be judged into profiles/ afterwards.
set -e
TAG=${1:-rX}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in fir fft sos resample; do
  steps=200; [ $w = fft ] && steps=30; [ $w = resample ] && steps=60
  python3 bench.py --workload $w --steps $steps --warmup 20 > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
  out=$ROOT/gpurun_out/${TAG}_prof_$w
  rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $ROOT/bench.py --workload $w --steps $steps --warmup 20 --no-cpu > $out.log 2>&1)
  cp $out/run_kernel_stats.csv gpurun_out/${TAG}_${w}_kernel_stats.csv
  PMC_TRAFFIC_ONLY=1 scripts/pmc_collect.sh ${TAG}_$w bench.py --workload $w --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1 || true
  echo "== $w"; tail -c 700 gpurun_out/${TAG}_bench_$w.json; echo
done
cat gpurun_out/pmc_${TAG}_*_summary.txt | grep -v "at::native\|rocclr\|hist_update" > gpurun_out/${TAG}_pmc_traffic.txt || true
cat gpurun_out/${TAG}_pmc_traffic.txt
