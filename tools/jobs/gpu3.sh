python tools/forward_ab.py --knob 1 --values 0,1 --rounds 5 > gpurun_out/r6_ab_twolevel.txt 2>&1
python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 4 > gpurun_out/r6_ab_vs_r5_b.txt 2>&1
python tools/g64_variants.py 5 > gpurun_out/r6_g64_variants.txt 2>&1
tail -3 gpurun_out/r6_ab_twolevel.txt; tail -5 gpurun_out/r6_ab_vs_r5_b.txt; cat gpurun_out/r6_g64_variants.txt
