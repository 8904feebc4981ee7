O=gpurun_out/r2i; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -s > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log | cut -c1-300
python tools/bench_ops.py --batch 16 --only sean > $O/ops_b16_sean.txt 2>&1; cat $O/ops_b16_sean.txt
python tools/bench_ops.py --batch 32 --only sean > $O/ops_b32_sean.txt 2>&1; cat $O/ops_b32_sean.txt
python tools/bench_ops_bf16.py --only sean,c1 > $O/ops_bf16.txt 2>&1; cat $O/ops_bf16.txt
