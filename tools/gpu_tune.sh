export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -2
python tools/bench_ops_bf16.py --only conv 2>&1 | grep "conv3x3" | tee $O/conv_bf16.txt
python tools/bench_ops_bf16.py --only conv --batch 16 --hw 128x160 2>&1 | grep "conv3x3" | tee -a $O/conv_bf16.txt
python bench.py --config c4 --steps 5 --warmup 2 | cut -c1-200
