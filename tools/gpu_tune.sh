export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -2
python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-200
python bench.py --config c4 --steps 5 --warmup 2 | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3s -o bench --output-format csv -- python3 bench.py --config c3 --steps 2 --warmup 1 --serial > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err; echo "prof c3 rc=$?"
python tools/kstats.py $O/prof_c3s 48 > $O/kstats_c3_serial.txt
