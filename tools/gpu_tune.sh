O=gpurun_out/r2s; mkdir -p $O
cd depth-aware-endoscopy-sr_amd && python - <<'PY'
import sys; sys.path.insert(0, "..")
import dasr_amd
from dasr_amd import build
build.build_hip(force=True, verbose=False, extra_flags=["-DDASR_CM_NOEPI"])
PY
cd ..
python tools/bench_ops.py --batch 16 --only conv 2>&1 | grep "conv3x3" | tee $O/noepi.txt
