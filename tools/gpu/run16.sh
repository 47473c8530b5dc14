set -x
mkdir -p gpurun_out
A2V_PARITY_REPORT=gpurun_out/parity_r03h.jsonl timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
