#!/bin/bash
# teacher forward on its side stream (default) vs on the main stream, whole steps, same box
B="python bench.py --steps 3 --warmup 1 --parity-leg skip --cpu-baseline skip --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2; do
  $B 2>/dev/null | python -c "$P" side-stream
  A2V_TEACHER_STREAM=0 $B 2>/dev/null | python -c "$P" main-stream
done
for gb in 16 8; do
  $B --global-batch $gb --steps 6 --warmup 2 2>/dev/null | python -c "$P" side-stream-gb$gb
  A2V_TEACHER_STREAM=0 $B --global-batch $gb --steps 6 --warmup 2 2>/dev/null | python -c "$P" main-stream-gb$gb
done
