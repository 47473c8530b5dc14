#!/bin/bash
# x3 conv via slab<float> (three launches) + 4-chunk split kernel: x3 tests, then the mode's speed and a kernel table
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "x3 or split or conv" > gpurun_out/t23_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/t23_rc.log
timeout 900 python -m pytest tests/test_pretrain_gpu.py -m gpu -x -q -k "large_full_size" > gpurun_out/t23_pre.log 2>&1
echo "pre rc=$?" >> gpurun_out/t23_rc.log
timeout 600 python tools/x3_bench.py > gpurun_out/t23_x3.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof23 -o x3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/x3_bench.py > $GRAFT_REPO_ROOT/gpurun_out/t23_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/t23_ops.log gpurun_out/t23_pre.log gpurun_out/t23_x3.log; cat gpurun_out/t23_rc.log
