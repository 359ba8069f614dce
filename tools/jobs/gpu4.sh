python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 4 > gpurun_out/r6_ab_vs_r5_c.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_c.txt
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 3 --batch 8 --n-blocks 4 --operands bf16 > gpurun_out/r6_ab_vs_r5_cfg2.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_cfg2.txt
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 3 --batch 1 > gpurun_out/r6_ab_vs_r5_b1.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_b1.txt
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r6_gpu_tests_b.txt 2>&1; tail -5 gpurun_out/r6_gpu_tests_b.txt
