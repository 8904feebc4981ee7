O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16" -s 2>&1 | grep -v "^$" | tail -6 | cut -c1-900
python tools/bench_ops_bf16.py --only sean > $O/sean.txt 2>&1; cat $O/sean.txt
for v in "4 2" "8 2" "8 4"; do set -- $v; echo "== NW=$1 MTW=$2"; DASR_CB_NW=$1 DASR_CB_MTW=$2 python tools/bench_ops_bf16.py --only conv 2>&1 | grep conv3x3 | tee -a $O/conv_variants.txt; done
