export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/gpu_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"; cat $O/bench.json
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
python tools/sean_split.py $O/prof 1 3 $O/sean_split.json > /dev/null; python tools/kstats.py $O/prof 40 > $O/kstats.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 tools/bench_ops.py --batch 32 --only sean --iters 3 > $O/pmc_fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 tools/bench_ops.py --batch 32 --only sean --iters 3 > $O/pmc_write.log 2>&1 && \
python tools/pmc_sean.py $O/pmc_fetch $O/pmc_write 32 > $O/pmc32.json; cp profiles/sean_fwd_pmc_b32.json $O/ 2>/dev/null
ls $O
