timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sean or blocks or depthnet or other_region or train" 2>&1 | tail -2
for i in 1 2; do python tools/bench_ops.py --batch 16 --only dynk 2>&1 | grep "dynk"; done
python tools/bench_ops.py --batch 32 --only dynk 2>&1 | grep "dynk"
