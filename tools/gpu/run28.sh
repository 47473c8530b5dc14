#!/bin/bash
# same-box A/B of two library builds, whole bench steps, alternating (new, prev, new, prev)
for i in 1 2; do
  python bench.py --steps 3 --warmup 1 --parity-leg skip --cpu-baseline skip --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('new ', d['value'], d['ms_per_step'])"
  A2V_LIB_PATH=$GRAFT_REPO_ROOT/animal2vec_amd/liba2v_hip_prev.so python bench.py --steps 3 --warmup 1 --parity-leg skip --cpu-baseline skip --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prev', d['value'], d['ms_per_step'])"
done
