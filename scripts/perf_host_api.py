#!/usr/bin/env python3
"""The headline configuration through the C++ drop-in API, by where the vectors live (pageable host memory,
page-locked host memory, resident): builds and runs tests/cpp/perf_host_api.cc and prints its JSON lines.
usage (GPU box): python3 scripts/perf_host_api.py [log2n]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(["make", "-C", os.path.join(ROOT, "libtsd_amd", "csrc"), "-s"], check=True)
subprocess.run(["make", "-C", os.path.join(ROOT, "libtsd_amd", "host"), "-s"], check=True)
subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "build/perf_host_api"], check=True)
sys.exit(subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "perf_host_api")] + sys.argv[1:]).returncode)
