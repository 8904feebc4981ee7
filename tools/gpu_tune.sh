O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv or bf16_ops or blocks or bf16_depthnet" 2>&1 | tail -3
python tools/bench_ops_bf16.py --only c1 2>&1 | grep mask | tee $O/c1.txt
python tools/bench_ops.py --only c1 2>&1 | grep mask | tee -a $O/c1.txt
python tools/bench_ops.py --only c1 --batch 32 2>&1 | grep mask | tee -a $O/c1.txt
