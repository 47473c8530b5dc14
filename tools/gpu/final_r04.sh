#!/bin/bash
# end-of-round evidence on one box: the GPU suite with the parity report, the bf16x3 mode's kernel stats, the default bench line
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
rm -f $out/r04_parity_report.jsonl
A2V_PARITY_REPORT=$out/r04_parity_report.jsonl timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/r04_final_gputest.log
bash tools/collect_profiles.sh r04b large_x3 "" "" tools/x3_bench.py > $out/r04_final_x3.log 2>&1
cd $root
timeout 600 python3 bench.py > $out/r04_final_bench.log 2>&1
grep '^{' $out/r04_final_bench.log > $out/r04_final_bench.json
tail -2 $out/r04_final_gputest.log
