# SQ counters of the bf16 convolution kernels (first kernel / persistent 8-wave) and of the split kernels (bf16 x 3, fp16 x 2): two passes each
export TMPDIR=/tmp
O=gpurun_out/${1:-r3pmc}; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES"
run() { # label, args...
  L=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P1 -d $O/${L}_a -o p --output-format csv -- python3 tools/bench_one_bf16.py "$@" > /dev/null 2>&1; echo "$L pass1 rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P2 -d $O/${L}_b -o p --output-format csv -- python3 tools/bench_one_bf16.py "$@" > /dev/null 2>&1; echo "$L pass2 rc=$?"
}
run first fwd 128 128 256 320 32 1
run v2 fwd 128 128 256 320 32 32
run split split 128 128 128 160 16
run splitw splitw 128 128 128 160 16
run split2 split2 128 128 128 160 16
run split2w split2w 128 128 128 160 16
python tools/pmc_conv.py $O/conv_sq_a.json first=$O/first_a v2=$O/v2_a split=$O/split_a splitw=$O/splitw_a split2=$O/split2_a split2w=$O/split2w_a > /dev/null
python tools/pmc_conv.py $O/conv_sq_b.json first=$O/first_b v2=$O/v2_b split=$O/split_b splitw=$O/splitw_b split2=$O/split2_b split2w=$O/split2w_b > /dev/null
python - <<PY
import json
a, b = json.load(open("$O/conv_sq_a.json")), json.load(open("$O/conv_sq_b.json"))
for run in a["runs"]:
    for k in a["runs"][run]:
        a["runs"][run][k].update({n: v for n, v in b["runs"].get(run, {}).get(k, {}).items() if n not in a["runs"][run][k]})
json.dump(a, open("$O/conv_sq_counters.json", "w"), indent=1)
for run, ks in a["runs"].items():
    for k, c in ks.items():
        print(run, k[:40], {n: c[n] for n in c if not n.startswith("SQ_")})
PY
