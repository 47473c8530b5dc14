#!/bin/bash
# round 6: launch-by-launch kernel trace of one step (micro-batch 64): the weight-gradient launches and their neighbours
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
export A2V_TEACHER_STREAM=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python3 $root/bench.py --steps 1 --warmup 1 --global-batch 64 --cpu-baseline skip --no-roofline --parity-leg skip > /tmp/tr1.log 2>&1 < /dev/null
tail -1 /tmp/tr1.log | cut -c1-120
f=$(find /tmp/tr1 -name "*kernel_trace.csv" | head -1)
head -1 $f
python3 $root/tools/trace_seq.py $f --match tn_bf16 --context 2 --limit 2000 > $out/r06_trace_tn.txt
python3 $root/tools/trace_seq.py $f --limit 100000 > $out/r06_trace_all.txt
wc -l $out/r06_trace_tn.txt $out/r06_trace_all.txt
