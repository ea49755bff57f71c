#!/bin/bash
# The developer / test switches (csrc/common.hpp: dev_switch) and the user switches of DESIGN.md select ALTERNATIVE PRODUCT PATHS that
# must stay parity-green: runs the test files that exercise each path with the switch set.  usage (GPU box, repo root): scripts/check_switches.sh
set -u
run() {   # run "ENV=1 [ENV2=..]" files...
  local envs=$1; shift
  local out
  out=$(env $envs python3 -m pytest "$@" -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1)
  echo "[$envs] $* -> $out"
  case "$out" in *failed*|*error*) FAILED=1;; esac
}
FAILED=0
run TSDGPU_FFT_GENERIC=1 tests/test_fft_gpu.py tests/test_ola_gpu.py
run TSDGPU_FFT_NO_SMOOTH=1 tests/test_fft_gpu.py
run TSDGPU_OLA_UNFUSED=1 tests/test_ola_gpu.py tests/test_detect_gpu.py
run TSDGPU_RFFT_TWO_PASS=1 tests/test_fft_gpu.py
run TSDGPU_POLY_COMPOSED=1 tests/test_polyphase_gpu.py -k "not direct_kernel"      # (that test asserts the bits of the fused / direct kernels)
run TSDGPU_POLY_NO_ROWS=1 tests/test_polyphase_gpu.py tests/test_host_pipeline_gpu.py
run TSDGPU_SHARD_SOS_HALO=1 tests/test_sharded_gpu.py -k "not long_memory"
run TSDGPU_SOS_NO_EXACT_CARRY=1 tests/test_sos_gpu.py tests/test_sharded_gpu.py -k "not long_memory"
# (the tests that assert WHICH path serves a filter, or its speed, are about the default choice)
run TSDGPU_RII_LITERAL=1 tests/test_polyphase_gpu.py -k "not block_parallel and not under_2ms and not under_1ms and not cliff and not literal_fallback and not complex_coefficients"
run TSDGPU_NO_BOUNCE=1 tests/test_fir_gpu.py tests/test_sos_gpu.py tests/test_fft_gpu.py
run TSDGPU_NO_PIPE=1 tests/test_host_pipeline_gpu.py
run TSDGPU_PIPE_ONE_THREAD=1 tests/test_host_pipeline_gpu.py
run TSDGPU_PIPE_CHUNK_MB=1 tests/test_host_pipeline_gpu.py
run TSDGPU_OLS_LONG_MIN=200 tests/test_fir_gpu.py
# round 3
run TSDGPU_OLS_DYN=0 tests/test_fir_gpu.py
run TSDGPU_OLS_RUN=1 tests/test_fir_gpu.py
run TSDGPU_RS_DYN=0 tests/test_resample_gpu.py
run TSDGPU_FFT_DYN=0 tests/test_fft_gpu.py
run TSDGPU_RS_LONG=0 tests/test_resample_gpu.py -k "not bit_for_bit"
run TSDGPU_RS_LONG_KMIN=8 tests/test_resample_gpu.py -k "not bit_for_bit"
run TSDGPU_RS15=0 tests/test_resample_gpu.py -k "not dynamic_tiles"
run TSDGPU_OLAW512=0 tests/test_ola_gpu.py
run TSDGPU_FFT_MIXED_UNFUSED=1 tests/test_fft_gpu.py tests/test_fft_sweep_gpu.py
run TSDGPU_FFT_NO_ODDPOW2=1 tests/test_fft_gpu.py tests/test_fft_sweep_gpu.py
run TSDGPU_FFT_ODDPOW2_ALL=1 tests/test_fft_gpu.py tests/test_fft_sweep_gpu.py
run TSDGPU_SHARD_NO_OVERLAP=1 tests/test_sharded_gpu.py
# round 4
run TSDGPU_SOS_FULL_SCAN=1 tests/test_sos_gpu.py tests/test_sharded_gpu.py
run "TSDGPU_RS_DYN_MIN=0 TSDGPU_OLS_DYN_MIN=0" tests/test_resample_gpu.py tests/test_fir_gpu.py
run TSDGPU_FFT_NO_2K=1 tests/test_fft_gpu.py tests/test_large_gpu.py
run TSDGPU_FFT_BLU_OLD=1 tests/test_fft_gpu.py tests/test_fft_sweep_gpu.py tests/test_ola_gpu.py
run TSDGPU_FFT_NO_3PASS=1 tests/test_fft_gpu.py tests/test_large_gpu.py
run TSDGPU_FFT_NO_1K_P1=1 tests/test_fft_gpu.py tests/test_fft_sweep_gpu.py
run TSDGPU_POLY_NO_DIRECT=1 tests/test_polyphase_gpu.py tests/test_resample_gpu.py -k "not direct_kernel"
exit $FAILED
