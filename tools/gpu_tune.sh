O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -2
echo "== TD bounce" | tee $O/conv_bf16_td.txt
python tools/bench_ops_bf16.py --only conv 2>&1 | grep "conv3x3" | tee -a $O/conv_bf16_td.txt
python tools/bench_ops_bf16.py --only conv --batch 16 --hw 128x160 2>&1 | grep "conv3x3" | head -2 | tee -a $O/conv_bf16_td.txt
cd depth-aware-endoscopy-sr_amd && python - <<'PY'
import sys, os; sys.path.insert(0, "..")
import dasr_amd
from dasr_amd import build
build.build_hip(force=True, verbose=False, extra_flags=["-DDASR_CB_NO_TD"])
PY
cd ..
echo "== no TD" | tee -a $O/conv_bf16_td.txt
python tools/bench_ops_bf16.py --only conv 2>&1 | grep "conv3x3" | tee -a $O/conv_bf16_td.txt
python tools/bench_ops_bf16.py --only conv --batch 16 --hw 128x160 2>&1 | grep "conv3x3" | head -2 | tee -a $O/conv_bf16_td.txt
