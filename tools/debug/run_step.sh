python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15
