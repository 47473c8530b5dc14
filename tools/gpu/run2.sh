set -x
mkdir -p gpurun_out
python -m pytest tests/test_masks_gpu.py -x -q 2>&1 | tail -5
python -m pytest tests/test_ops_gpu.py -x -q -k "attention or alibi or streams" 2>&1 | tail -15
A2V_PARITY_REPORT=gpurun_out/parity_r03a.jsonl python -m pytest tests/test_pretrain_gpu.py -q -k "large_full_size or long_sequence or base_star_pretrain or hd64 or finetune_large or config5" 2>&1 | tail -15
