O=gpurun_out/r2m; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -s > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log | cut -c1-200
python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2.json
python bench.py --config c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
