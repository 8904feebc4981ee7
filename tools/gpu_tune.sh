O=gpurun_out/r2q; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16" 2>&1 | tail -2
python tools/bench_ops_bf16.py --only conv 2>&1 | grep "conv3x3" | tee $O/conv_ep.txt
python bench.py --config c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
