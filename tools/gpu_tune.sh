O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sean or bf16_ops or blocks or depthnet or other_region or soft" 2>&1 | tail -2
for F in "-DSB_DEEP_PREFETCH=1" "-DSB_DEEP_PREFETCH=0" "-DSB_DEEP_PREFETCH=1"; do
cd depth-aware-endoscopy-sr_amd && FLAGS="$F" python - <<'PY'
import sys, os; sys.path.insert(0, "..")
import dasr_amd
from dasr_amd import build
build.build_hip(force=True, verbose=False, extra_flags=os.environ["FLAGS"].split())
PY
cd ..
echo "== $F" | tee -a $O/sean_bwd.txt
python tools/bench_ops.py --batch 16 --only sean 2>&1 | grep "sean_bwd" | tee -a $O/sean_bwd.txt
python tools/bench_ops_bf16.py --only sean 2>&1 | grep "sean_bwd" | tee -a $O/sean_bwd.txt
python tools/bench_ops_bf16.py --only sean 2>&1 | grep "sean_bwd" | tee -a $O/sean_bwd.txt
python tools/bench_ops_bf16.py --only sean --batch 16 --hw 128x160 2>&1 | grep "sean_bwd" | tee -a $O/sean_bwd.txt
done
