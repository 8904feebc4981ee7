python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-160
python bench.py --config c3 --steps 3 --warmup 1 --wgrad-stream | cut -c1-160
python bench.py --config c4 --steps 5 --warmup 2 | cut -c1-160
python bench.py --config c4 --steps 5 --warmup 2 --wgrad-stream | cut -c1-160
