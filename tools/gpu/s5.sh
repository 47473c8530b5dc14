#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
rm -f $out/r05_parity_report.jsonl
A2V_PARITY_REPORT=$out/r05_parity_report.jsonl timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $out/s5_gputest.txt
cat $out/s5_gputest.txt
