python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or attention" 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "(matches_oracle and not f32) or q1 or trained" 2>&1 | grep -v amdgpu.ids | tail -3
{ echo "# same-box A/B of two builds (tools/ab_build.sh): default (k = 3 convolution: 256 x 128 workgroup tile, 128 x 64 per wave) vs -DEV2H_NO_TAP3_WIDE (128 x 128, 64 x 64 per wave); windows/s, ms/step, HIP-event ms of the dominant launch site";
echo "## f16x2, B=128, N=8192 (the dominant site is the k=3 GEMM itself)"; AB_ARGS="--points 8192 --batch 128 --steps 40 --warmup 5 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-host-io" bash tools/ab_build.sh 2 "" "-DEV2H_NO_TAP3_WIDE";
echo "## f16x2, B=256, N=2048"; bash tools/ab_build.sh 2 "" "-DEV2H_NO_TAP3_WIDE";
echo "## bf16, B=256, N=2048"; AB_ARGS="--precision bf16 --steps 100 --warmup 5 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-host-io" bash tools/ab_build.sh 2 "" "-DEV2H_NO_TAP3_WIDE";
python -m ev2hands_amd.build --force > /dev/null 2>&1; } > gpurun_out/r5_ab_tap3_wide.txt 2>&1
cat gpurun_out/r5_ab_tap3_wide.txt
