O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" -s 2>&1 | grep -v "^$" | tail -22 | cut -c1-420
python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-200
python bench.py --config c4 --steps 5 --warmup 2 | cut -c1-200
