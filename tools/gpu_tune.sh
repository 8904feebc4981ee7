timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "blocks or depthnet or bf16_ops or batch" 2>&1 | tail -2
for i in 1 2 3; do python tools/bench_ops.py --batch 16 --only sean 2>&1 | grep "instnorm"; done
python tools/bench_ops.py --batch 32 --only sean 2>&1 | grep "instnorm"
python tools/bench_ops_bf16.py --only sean 2>&1 | grep "instnorm"
