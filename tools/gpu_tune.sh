export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O; rm -rf $O/pmcA $O/pmcB
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/pmcA -o p --output-format csv -- python3 tools/bench_one_bf16.py fwd > /dev/null 2> $O/pmcA.err; echo "A rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmcB -o p --output-format csv -- python3 tools/bench_one_bf16.py fwd > /dev/null 2> $O/pmcB.err; echo "B rc=$?"
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r2s/pmcA", "gpurun_out/r2s/pmcB"):
    agg = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_conv3x3_bf16" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(d[-4:], k, sum(v) / len(v), len(v))
PY
