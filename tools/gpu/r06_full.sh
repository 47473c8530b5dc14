#!/bin/bash
# round 6: whole GPU test-suite, then the default bench
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06_gpu_tests.txt
tail -4 gpurun_out/r06_gpu_tests.txt
python3 bench.py --cpu-baseline skip --parity-leg skip > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err; tail -1 gpurun_out/r06_bench_a.json | cut -c1-600
A2V_DEBUG_SET=39=0 python3 bench.py --cpu-baseline skip --parity-leg skip --no-roofline > gpurun_out/r06_bench_old.json 2> gpurun_out/r06_bench_old.err; tail -1 gpurun_out/r06_bench_old.json | cut -c1-300
python3 bench.py --cpu-baseline skip --parity-leg skip --no-roofline > gpurun_out/r06_bench_b.json 2>/dev/null; tail -1 gpurun_out/r06_bench_b.json | cut -c1-300
