for t in default 64 32 16; do
  if [ $t = default ]; then unset OMP_NUM_THREADS; else export OMP_NUM_THREADS=$t; fi
  echo "threads=$t"; timeout 600 python -m pytest tests/test_gpu_groups_n256.py -q -k "stage_4" 2>&1 | tail -1
done
