python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-b32 2>&1 | grep "value\|enqueue" | cut -c1-330
python bench.py --config c3 --steps 20 --warmup 2 2>&1 | grep "value\|enqueue" | cut -c1-200
python bench.py --config c4 --steps 60 --warmup 2 2>&1 | grep "value\|enqueue" | cut -c1-200
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
