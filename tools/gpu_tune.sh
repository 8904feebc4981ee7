O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sean or bf16_ops or blocks or depthnet or other_region or soft" 2>&1 | tail -2
python tools/bench_ops.py --batch 16 --only sean 2>&1 | grep "sean_bwd" 
python tools/bench_ops.py --batch 32 --only sean 2>&1 | grep "sean_bwd"
python tools/bench_ops_bf16.py --only sean 2>&1 | grep "sean_bwd"
python tools/bench_ops_bf16.py --only sean 2>&1 | grep "sean_bwd"
python tools/bench_ops_bf16.py --only sean --batch 16 --hw 128x160 2>&1 | grep "sean_bwd"
