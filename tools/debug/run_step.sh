python -m pytest tests/test_gpu_ops.py tests/test_gpu_stress.py -x -q -m gpu -k "ball or stress or lattice or select" 2>&1 | tail -3
python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "matches_oracle and f16x2" 2>&1 | tail -3
bash tools/n8192_sweep.sh r5 _ballxcd
bash tools/profile_n8192.sh r5
