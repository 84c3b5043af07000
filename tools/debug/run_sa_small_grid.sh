AMAX=1 python tools/debug/sa_small_grid.py > gpurun_out/r5_sa_small_grid_amax.txt 2>&1
tail -50 gpurun_out/r5_sa_small_grid_amax.txt
