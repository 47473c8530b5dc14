#!/bin/bash
# full GPU suite + smoke + a 20-step bench (memory / rate stability)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/t31_gpu.log 2>&1
echo "gpu rc=$?" > gpurun_out/t31_rc.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/t31_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/t31_rc.log
A2V_BENCH_STEP_TIMES=0 timeout 1200 python bench.py --steps 20 --warmup 2 --cpu-baseline skip --parity-leg skip --no-roofline > gpurun_out/t31_bench20.json 2> gpurun_out/t31_bench20.err
echo "bench rc=$?" >> gpurun_out/t31_rc.log
tail -n 3 gpurun_out/t31_gpu.log; tail -n 1 gpurun_out/t31_smoke.log; cat gpurun_out/t31_rc.log; head -c 260 gpurun_out/t31_bench20.json
