#!/bin/bash
# GPU helper: weight-gradient time per shape for forced pixel-split counts (asm_tuning.wgrad_splits) vs the cost model's choice
for shape in C64-K256-1x1-H28 C128-K512-1x1-H14 C512-K2048-1x1-H7 C128-K256-1x1-H56 C256-K1024-1x1-H7 C256-K256-1x1-H28 C1024-K1024-1x1-H14 C1024-K256-1x1-H7 C256-K64-1x1-H56 C512-K512-1x1-H14 C256-K512-1x1-H28 C2048-K512-1x1-H7 C512-K1024-3x3-H14 C256-K512-3x3-H7 C128-K256-3x3-H14 C512-K1024-3x3-H7 C256-K512-3x3-H14 C128-K256-3x3-H28 C64-K128-3x3-H56; do
  line="$shape"
  for sp in 0 2 4 8 16 32 64 128 256; do
    t=$(ASM_WGRAD_SPLITS=$sp python tools/conv_bench.py --kinds wgrad --only $shape --top 1 --iters 20 2>/dev/null | grep "^x" | sed -E 's/.*\| +([0-9.]+) us.*/\1/')
    line="$line  sp$sp=$t"
  done
  echo "$line"
done
