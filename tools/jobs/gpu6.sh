python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "w1 or g64" > gpurun_out/r6_gpu_w1_tests.txt 2>&1; tail -3 gpurun_out/r6_gpu_w1_tests.txt
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 4 > gpurun_out/r6_ab_vs_r5_e.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_e.txt
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 3 --batch 1 > gpurun_out/r6_ab_vs_r5_b1_e.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_b1_e.txt
python -m pytest tests/test_gpu_forward.py tests/test_gpu_seed_sweep.py -m gpu -x -q > gpurun_out/r6_gpu_fwd_tests.txt 2>&1; tail -3 gpurun_out/r6_gpu_fwd_tests.txt
