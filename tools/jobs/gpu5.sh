python tools/forward_ab.py --knob 2 --values 0,1 --rounds 5 > gpurun_out/r6_ab_groups.txt 2>&1
tail -3 gpurun_out/r6_ab_groups.txt
python tools/forward_ab.py --knob 2 --values 0,1 --rounds 5 --batch 8 --n-blocks 4 > gpurun_out/r6_ab_groups_b8.txt 2>&1
tail -3 gpurun_out/r6_ab_groups_b8.txt
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 4 > gpurun_out/r6_ab_vs_r5_d.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_d.txt
python -m pytest tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r6_gpu_fwd_tests.txt 2>&1; tail -3 gpurun_out/r6_gpu_fwd_tests.txt
