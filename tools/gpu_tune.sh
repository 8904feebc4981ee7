python tools/prof_small_ops.py bf16 2>&1 | grep -v amdgpu.ids | head -45
