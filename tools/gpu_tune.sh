export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-200
python bench.py --config c4 --steps 5 --warmup 2 | cut -c1-200
