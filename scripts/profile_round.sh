#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <round-tag>
# Produces under gpurun_out/: rocprofv3 --kernel-trace --stats summaries of the bench command of every
# workload, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only) and, LAST, the bench
# JSON lines themselves -- after the PMC summary has been put under profiles/ on the box, so that the
# lines read their `roofline.traffic` from this round's counters.  Copy what is to be judged into
# profiles/ afterwards (the same file names).
set -e
TAG=${1:-rX}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
steps_of() { case $1 in fft) echo 30;; resample) echo 60;; *) echo 200;; esac; }
for w in fir fft sos resample; do
  steps=$(steps_of $w)
  out=$ROOT/gpurun_out/${TAG}_prof_$w
  rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $ROOT/bench.py --workload $w --steps $steps --warmup 20 --no-cpu > $out.log 2>&1)
  cp $out/run_kernel_stats.csv gpurun_out/${TAG}_${w}_kernel_stats.csv
  PMC_TRAFFIC_ONLY=1 scripts/pmc_collect.sh ${TAG}_$w bench.py --workload $w --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1 || true
  echo "== $w profiled"
done
cat gpurun_out/pmc_${TAG}_*_summary.txt | grep -v "at::native\|rocclr\|hist_update" > gpurun_out/${TAG}_pmc_traffic.txt || true
cp gpurun_out/${TAG}_pmc_traffic.txt profiles/${TAG}_pmc_traffic.txt
cat gpurun_out/${TAG}_pmc_traffic.txt
for w in fir fft sos resample; do
  steps=$(steps_of $w)
  python3 bench.py --workload $w --steps $steps --warmup 20 > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
  echo "== $w"; tail -c 900 gpurun_out/${TAG}_bench_$w.json; echo
done
# the secondary tables of the round (a change tuned on the headline configuration is re-measured where the kernel is actually used:
# round 4's direct-FIR stores cost nothing at 127 taps and a fifth at 7 ... 31)
python3 scripts/perf_short_fir.py > gpurun_out/${TAG}_perf_short_fir.txt 2>&1 || true
python3 scripts/perf_fft_sizes.py > gpurun_out/${TAG}_perf_fft_sizes.txt 2>&1 || true
python3 scripts/perf_fft_sizes.py 125 131 250 257 375 511 513 1001 2000 3000 4000 6000 8190 >> gpurun_out/${TAG}_perf_fft_sizes.txt 2>&1 || true
python3 scripts/perf_fft_sizes.py --total 28 1048576 2097152 4194304 8388608 16777216 33554432 65536 262144 1024 4096 > gpurun_out/${TAG}_perf_fft_sizes_2p28.txt 2>&1 || true
python3 scripts/perf_welch.py 125 250 375 1000 1001 2000 3000 256 1024 4096 > gpurun_out/${TAG}_perf_welch.txt 2>&1 || true
python3 scripts/perf_secondary.py > gpurun_out/${TAG}_perf_secondary.txt 2>&1 || true
python3 scripts/perf_polyphase.py > gpurun_out/${TAG}_perf_polyphase.txt 2>&1 || true
echo "== secondary tables written"
