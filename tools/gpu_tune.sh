O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv or blocks or depthnet or train" 2>&1 | tail -2
for i in 1 2; do
echo "== old"; DASR_HIPEMU_LIB=$PWD/depth-aware-endoscopy-sr_amd/libdasr_hip_old.so python tools/bench_ops.py --batch 16 --only conv 2>&1 | grep "conv3x3"
echo "== new"; python tools/bench_ops.py --batch 16 --only conv 2>&1 | grep "conv3x3"
done
echo "== old c2"; DASR_HIPEMU_LIB=$PWD/depth-aware-endoscopy-sr_amd/libdasr_hip_old.so python bench.py --no-cpu-baseline --no-b32 --steps 3 --warmup 1 | cut -c1-160
echo "== new c2"; python bench.py --no-cpu-baseline --no-b32 --steps 3 --warmup 1 | cut -c1-160
echo "== old c2"; DASR_HIPEMU_LIB=$PWD/depth-aware-endoscopy-sr_amd/libdasr_hip_old.so python bench.py --no-cpu-baseline --no-b32 --steps 3 --warmup 1 | cut -c1-160
echo "== new c2"; python bench.py --no-cpu-baseline --no-b32 --steps 3 --warmup 1 | cut -c1-160
