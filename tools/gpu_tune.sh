O=gpurun_out/r2s; mkdir -p $O
for i in 1 2; do
echo "== old"; DASR_HIPEMU_LIB=$PWD/depth-aware-endoscopy-sr_amd/libdasr_hip_old.so python tools/bench_ops_bf16.py --only conv 2>&1 | grep "conv3x3" | head -2
echo "== new"; python tools/bench_ops_bf16.py --only conv 2>&1 | grep "conv3x3" | head -2
done
echo "== old c3"; DASR_HIPEMU_LIB=$PWD/depth-aware-endoscopy-sr_amd/libdasr_hip_old.so python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-160
echo "== new c3"; python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-160
echo "== old c3"; DASR_HIPEMU_LIB=$PWD/depth-aware-endoscopy-sr_amd/libdasr_hip_old.so python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-160
echo "== new c3"; python bench.py --config c3 --steps 3 --warmup 1 | cut -c1-160
