#!/usr/bin/env python3
"""Split the k_sean_fwd_onehot dispatches of a `rocprofv3 --kernel-trace -- python3 bench.py ...` run by the phase of
bench.py they belong to, so that profiles/ can reproduce bench.py's roofline numbers from tracked files:

  phase A  warm-up + timed steps      26 launches per step, co-running with the side-stream convolutions (B=16)
  phase B  one un-overlapped step     26 launches (bench.py `roofline`)
  phase C  batch-32 forward-only      26 launches x (1 warm-up + 3 timed passes) (bench.py `roofline_b32`)

Usage: python tools/sean_split.py <rocprof output dir> <warmup> <steps> [out.json]"""
import csv
import glob
import json
import os
import sys

d, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
csvs = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1:]
if not csvs:
    sys.exit("no kernel trace csv under " + d)
rows = [r for r in csv.DictReader(open(csvs[0])) if "k_sean_fwd_onehot" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def has_residual(name):
    """k_sean_fwd_onehot<RELU, HAS_RES[, T]>: the second template argument."""
    args = name.split("<", 1)[1].split(">", 1)[0].split(",")
    return args[1].strip() in ("true", "1")


dur = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, has_residual(r["Kernel_Name"])) for r in rows]
nA = 26 * (warmup + steps)
phases = {"overlapped_steps_b16": dur[:nA], "unoverlapped_step_b16": dur[nA:nA + 26],
          "forward_only_b32": dur[nA + 26 + 26:nA + 26 + 26 * 4]}      # first B=32 pass is the warm-up
PX, C, K = 128 * 160, 64, 10


def summarise(sel, B):
    out = {}
    for res in (False, True):
        t = [x for x, r in sel if r == res]
        if not t:
            continue
        us = sum(t) / len(t)
        alg = (4 * (4 * C + (C if res else 0)) + 4 * K) * B * PX
        true_b = (4 * (4 * C + (C if res else 0)) + 1) * B * PX
        out["residual" if res else "no_residual"] = {
            "launches": len(t), "avg_us": round(us, 2), "min_us": round(min(t), 2),
            "frac_of_8TBs_survey_8d_bytes": round(alg / us / 1e3 / 8000.0, 4),        # bytes / us / 1e3 = GB/s
            "frac_of_8TBs_kernel_minimum_bytes": round(true_b / us / 1e3 / 8000.0, 4)}
    if sel:
        out["avg_us_all"] = round(sum(x for x, _ in sel) / len(sel), 2)
    return out


res = {"source": "rocprofv3 --kernel-trace -- python3 bench.py --warmup %d --steps %d (%s)" % (warmup, steps, os.path.basename(csvs[0])),
       "kernel": "k_sean_fwd_onehot", "dispatches": len(dur)}
for name, sel in phases.items():
    res[name] = summarise(sel, 32 if name.endswith("b32") else 16)
txt = json.dumps(res, indent=1)
print(txt)
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write(txt + "\n")
