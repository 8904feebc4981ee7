O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -s -k "bf16" > $O/t.log 2>&1; grep "conv9\|passed\|failed\|Error" $O/t.log | cut -c1-600
python tools/bench_ops_bf16.py --only conv9 > $O/conv9.txt 2>&1; cat $O/conv9.txt
python tools/bench_ops_bf16.py --only conv9 --batch 16 --hw 128x160 >> $O/conv9.txt 2>&1; tail -1 $O/conv9.txt
python bench.py --config c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
