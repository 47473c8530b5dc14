#!/bin/bash
# round 6: whole GPU test-suite (with the parity report), smoke(), then the default bench line (with its CPU-baseline and parity legs)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
rm -f gpurun_out/r06_parity_report.jsonl
A2V_PARITY_REPORT=$root/gpurun_out/r06_parity_report.jsonl python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06_gpu_tests.txt
tail -4 gpurun_out/r06_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.txt 2>&1; tail -2 gpurun_out/r06_smoke.txt
python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -1 gpurun_out/r06_bench_default.json | cut -c1-600
