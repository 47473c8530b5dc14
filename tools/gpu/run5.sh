set -x
mkdir -p gpurun_out
A2V_PARITY_REPORT=gpurun_out/parity_r03d.jsonl python -m pytest tests/test_pretrain_gpu.py -q -x -k "vs_oracle or config5" 2>&1 | tail -5
A2V_DEBUG_SET=18=1 A2V_PARITY_REPORT=gpurun_out/parity_r03_msum.jsonl python -m pytest tests/test_pretrain_gpu.py -q -k "long_sequence or large_full_size or base_star_pretrain" 2>&1 | tail -3
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 --debug-set 18=1 > gpurun_out/b5_msum.json 2>&1; tail -c 200 gpurun_out/b5_msum.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b5_nomsum.json 2>&1; tail -c 200 gpurun_out/b5_nomsum.json
bash tools/gpu/run_profiles.sh r03 2>&1 | tail -40
