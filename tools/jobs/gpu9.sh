python tools/forward_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_r5.so --reps 4 > gpurun_out/r6_ab_vs_r5_f.txt 2>&1
tail -5 gpurun_out/r6_ab_vs_r5_f.txt
python bench.py > gpurun_out/r6_bench_b.json 2> gpurun_out/r6_bench_b.err; python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['max_abs_delta_vs_oracle'], d['parity_gate'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['host'], d['secondary_bf16_cfg2']['value'])"
