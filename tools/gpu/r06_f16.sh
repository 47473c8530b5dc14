#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
for f in 1 0; do
  A2V_TEACHER_F16=$f A2V_PARITY_REPORT=gpurun_out/r06_f16_report_$f.jsonl python3 -m pytest tests/test_pretrain_gpu.py -x -q -m gpu -k "(trained_golden or large_full_size_vs_oracle or test_hd64_vs_oracle) and dtype1" 2>&1 | tail -4
done
python3 - <<'PY'
import json
for f in (1,0):
    for l in open(f"gpurun_out/r06_f16_report_{f}.jsonl"):
        r=json.loads(l)
        print("teacher_f16" if f else "teacher_bf16", r["test"], "targets", r.get("targets"), "median", r.get("median"), "worst", r["worst"][0][0])
PY
