O=gpurun_out/r2h; mkdir -p $O
python tools/bench_ops.py --batch 16 > $O/ops_b16.txt 2>&1; cat $O/ops_b16.txt
python tools/bench_ops.py --batch 32 --only sean,dynk > $O/ops_b32_sean.txt 2>&1; cat $O/ops_b32_sean.txt
python tools/bench_ops_bf16.py > $O/ops_bf16_c3.txt 2>&1; cat $O/ops_bf16_c3.txt
python bench.py --batch 32 --no-cpu-baseline --no-b32 > $O/bench_batch32.json 2> $O/bench_batch32.err; cat $O/bench_batch32.json | cut -c1-400
