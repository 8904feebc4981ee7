#!/usr/bin/env python3
"""SQ counters of the trunk convolution kernels from rocprofv3 --pmc runs (CSV) of tools/bench_one_bf16.py: per kernel,
averages over its dispatches, plus the derived figures DESIGN.md quotes (MFMA-busy share, VALU instructions per MFMA, LDS
bank-conflict share).  Usage: python tools/pmc_conv.py <out.json> <label>=<rocprof dir> [...]"""
import collections
import csv
import glob
import json
import sys

out = {"source": "rocprofv3 --pmc (SQ passes; --kernel-trace only) over tools/bench_one_bf16.py on MI355X; counter values are "
                 "sums over the chip per dispatch, averaged over the dispatches of the run", "runs": {}}
for spec in sys.argv[2:]:
    label, d = spec.split("=", 1)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "conv3x3" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)                 # kernel durations of the same run (the --kernel-trace CSV)
    for path in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "conv3x3" in k:
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    run = {}
    for k, ctrs in agg.items():
        c = {n: sum(v) / len(v) for n, v in ctrs.items()}
        d_ = dict(c)
        if dur.get(k):
            d_["avg_duration_us"] = round(sum(dur[k]) / len(dur[k]) / 1e3, 2)
            if c.get("SQ_BUSY_CU_CYCLES"):
                # every CU is busy for the whole launch of these persistent kernels: busy cycles per CU / duration = shader clock
                d_["shader_clock_ghz_from_busy_cycles"] = round(c["SQ_BUSY_CU_CYCLES"] / 256 / (sum(dur[k]) / len(dur[k])), 3)
        if c.get("SQ_BUSY_CU_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d_["mfma_busy_share_of_cu_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"] / 4, 4)
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_INSTS_VALU"):
            d_["valu_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 3)
        if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_LDS_BANK_CONFLICT") is not None:
            d_["lds_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_WAVE_CYCLES"):
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c.get(n):
                    d_[n.lower() + "_share_of_wave_cycles"] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
        run[k] = d_
    out["runs"][label] = run
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
