python tools/g64_variants.py 5 2>&1 | head -12 > gpurun_out/r6_s32_variants.txt; cat gpurun_out/r6_s32_variants.txt
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "h2s32 or g64" > gpurun_out/r6_gpu_s32_tests.txt 2>&1; tail -3 gpurun_out/r6_gpu_s32_tests.txt
python bench.py --config cfg3 --steps 50 --warmup 10 > gpurun_out/r6_bench_cfg3_a.json 2> gpurun_out/r6_bench_cfg3_a.err; python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_cfg3_a.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['max_abs_delta_vs_oracle'], d['parity_gate'])"
