python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "upsample" > gpurun_out/r6_gpu_up_tests.txt 2>&1; tail -2 gpurun_out/r6_gpu_up_tests.txt
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 4 > gpurun_out/r6_ab_vs_r5_g.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_g.txt
python -m pytest tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r6_gpu_fwd_tests2.txt 2>&1; tail -2 gpurun_out/r6_gpu_fwd_tests2.txt
