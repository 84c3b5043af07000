python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "pose or fixture or matches_oracle or two_stream or config_sizes" 2>&1 | grep -v amdgpu.ids | tail -6
python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py tests/test_gpu_range.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --steps 100 --no-legs --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/r5_line_c.json
python -c "
import json; d=json.load(open('gpurun_out/r5_line_c.json')); print('default:', d['value'], d['ms_per_step'], d.get('latency_ms'))"
