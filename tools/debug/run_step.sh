python -m pytest tests/test_gpu_range.py -x -q -m gpu -k "exact_window_maxima or trained_checkpoint_under" 2>&1 | grep -v amdgpu.ids | tail -6
for i in 1 2 3; do python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "config5" 2>&1 | grep -v amdgpu.ids | tail -1; done
