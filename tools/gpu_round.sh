# one GPU-box visit: GPU tests (log kept), fp32 bench, bf16 benches, rocprof of both; outputs under gpurun_out/$1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/gpu_tests.log | cut -c1-200
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"; cut -c1-300 $O/bench.json
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-split --no-cpu-baseline --no-b32 > $O/bench_nosplit.json 2> $O/bench_nosplit.err; echo "bench nosplit rc=$?"; cut -c1-200 $O/bench_nosplit.json
timeout -k 10 400 python bench.py --config c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; rc=$?; echo "bench c3 rc=$rc"; cut -c1-300 $O/bench_c3.json
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --config c4 --steps 10 --warmup 3 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"; cut -c1-200 $O/bench_c4.json; tail -1 $O/bench_c4.err
timeout -k 10 300 python bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"; cut -c1-200 $O/bench_c5.json
timeout -k 10 200 python bench.py --batch 32 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_b32.json 2> $O/bench_b32.err; echo "bench b32 rc=$?"; cut -c1-200 $O/bench_b32.json
timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline --no-b32 --steps 10 --warmup 2 > $O/bench_infer_c2.json 2> $O/bench_infer_c2.err; echo "infer c2 rc=$?"; cut -c1-200 $O/bench_infer_c2.json
timeout -k 10 200 python bench.py --mode infer --config c4 --steps 10 --warmup 2 > $O/bench_infer_c4.json 2> $O/bench_infer_c4.err; echo "infer c4 rc=$?"; cut -c1-200 $O/bench_infer_c4.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
python tools/sean_split.py $O/prof 1 3 $O/sean_split.json > /dev/null; python tools/kstats.py $O/prof 45 > $O/kstats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c2s -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-b32 > $O/bench_c2_serial.json 2> $O/bench_c2_serial.err; echo "prof c2 serial rc=$?"
python tools/kstats.py $O/prof_c2s 60 > $O/kstats_c2_serial.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3s -o bench --output-format csv -- python3 bench.py --config c3 --steps 2 --warmup 1 --serial > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err; echo "prof c3 serial rc=$?"
python tools/kstats.py $O/prof_c3s 60 > $O/kstats_c3_serial.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4s -o bench --output-format csv -- python3 bench.py --config c4 --steps 3 --warmup 1 --serial > $O/bench_c4_serial.json 2> $O/bench_c4_serial.err; echo "prof c4 serial rc=$?"
python tools/kstats.py $O/prof_c4s 60 > $O/kstats_c4_serial.txt
timeout -k 10 200 python tools/bench_ops.py --batch 16 > $O/ops_b16.txt 2>&1; echo "ops rc=$?"
timeout -k 10 200 python tools/bench_ops.py --batch 32 --only sean,c1 > $O/ops_b32.txt 2>&1
timeout -k 10 300 python tools/bench_ops_bf16.py > $O/ops_bf16_c3.txt 2>&1; echo "ops bf16 rc=$?"
timeout -k 10 200 python tools/bench_ops_bf16.py --batch 16 --hw 128x160 > $O/ops_bf16_c4.txt 2>&1
timeout -k 10 300 python tools/bench_split.py 16 > $O/ops_split_b16.txt 2>&1; timeout -k 10 300 python tools/bench_split.py 32 > $O/ops_split_b32.txt 2>&1
timeout -k 10 200 python tools/bench_conv9.py 16 > $O/ops_conv9_b16.txt 2>&1; timeout -k 10 200 python tools/bench_amax.py > $O/ops_amax_b16.txt 2>&1
timeout -k 10 300 python bench.py --split-pieces 3 --no-cpu-baseline --no-b32 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; echo "bench bf16x3 rc=$?"; cut -c1-200 $O/bench_bf16x3.json
ls $O
