set -x
mkdir -p gpurun_out
for gb in 128 64 32 16 8; do
  st=4; [ $gb -le 16 ] && st=8
  python bench.py --steps $st --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch $gb > gpurun_out/r03_sweep_gb$gb.json 2>&1
  tail -c 150 gpurun_out/r03_sweep_gb$gb.json
done
python bench.py --model base --global-batch 64 --steps 3 --warmup 1 --cpu-baseline skip --parity-leg skip > gpurun_out/r03_bench_base.json 2>&1; tail -c 400 gpurun_out/r03_bench_base.json
python bench.py --model large44k --samples 441000 --global-batch 4 --micro-batch 2 --steps 2 --warmup 1 --cpu-baseline skip --parity-leg skip > gpurun_out/r03_bench_large44k.json 2>&1; tail -c 400 gpurun_out/r03_bench_large44k.json
python tools/finetune_bench.py --steps 2 --warmup 1 --accum 2 > gpurun_out/r03_finetune.txt 2>&1; cat gpurun_out/r03_finetune.txt | tail -4
