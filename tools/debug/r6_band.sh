# the row-band form of wgrad_halo_kernel (56- and 28-wide maps, K = 128 as two halves) against the gather kernels (ASM_WGRAD_HALO=0)
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "wgrad" 2>&1 | tail -4
SH="--shape 256,56,56,64,128,3,3,1 --shape 256,28,28,64,128,3,3,1 --shape 256,56,56,32,64,3,3,1 --shape 256,112,112,64,32,3,3,1 --shape 256,112,112,32,64,3,3,1 --shape 256,112,112,32,32,3,3,1"
for rep in 1 2; do
  echo "== halo"; timeout 300 python tools/conv_bench.py --kinds wgrad --iters 20 $SH 2>&1 | tail -8
  echo "== gather"; ASM_WGRAD_HALO=0 timeout 300 python tools/conv_bench.py --kinds wgrad --iters 20 $SH 2>&1 | tail -8
done
