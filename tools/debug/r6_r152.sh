timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "r152_forward" 2>&1 | grep -E "^E |passed|failed" | head
