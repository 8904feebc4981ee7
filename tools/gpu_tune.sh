export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv or blocks or depthnet or encoder" 2>&1 | tail -2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4s -o bench --output-format csv -- python3 bench.py --config c4 --steps 3 --warmup 1 --serial > $O/bench_c4_serial.json 2> $O/bench_c4_serial.err; echo "prof c4 rc=$?"
python tools/kstats.py $O/prof_c4s 45 > $O/kstats_c4_serial.txt
cat $O/bench_c4_serial.json | cut -c1-200
python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-200
python bench.py --config c4 --steps 3 --warmup 1 | cut -c1-200
