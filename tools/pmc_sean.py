#!/usr/bin/env python3
"""HBM bytes per launch of k_sean_fwd_onehot from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; --output-format
csv) over `tools/bench_ops.py --batch B --only sean --amax` (the instantiation the fp32 step runs: it also keeps the running
max |out|) -> profiles/sean_fwd_pmc.json (read by bench.py for roofline.traffic).
Usage: python tools/pmc_sean.py <fetch dir> <write dir> [B]   (B != 16 writes profiles/sean_fwd_pmc_b<B>.json)"""
import collections
import csv
import glob
import json
import os
import sys


def load(d, ctr):
    agg = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr and "k_sean_fwd_onehot" in r["Kernel_Name"]:
                targs = r["Kernel_Name"].split("(")[0].split("<", 1)[1].rsplit(">", 1)[0].split(",")   # <RELU, HAS_RES, T, AMAX>
                agg["res" if targs[1].strip() in ("true", "1") else "nores"].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
H, W, C, K = 128, 160, 64, 10
px = B * H * W
hbm = {k: 2 * fetch[k] * 1024 + write[k] * 1024 for k in fetch}
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_ops.py --batch %d "
              "--only sean --amax, kernel k_sean_fwd_onehot<relu, residual, float, amax>, MI355X" % B,
    "units_note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) "
                  "coalesced streaming read (MI355X_MICROARCH.md, HBM), so fetch bytes = 2 * FETCH_SIZE * 1024; "
                  "WRITE_SIZE is exact (round 2's residual variant wrote 7 MB more: register spills, gone since it runs two workgroups per CU)",
    "B": B, "H": H, "W": W, "C": C, "K": K,
    "fetch_kib_no_residual": fetch["nores"], "fetch_kib_residual": fetch["res"],
    "write_kib_no_residual": write["nores"], "write_kib_residual": write["res"],
    "hbm_bytes_per_launch_no_residual": hbm["nores"], "hbm_bytes_per_launch_residual": hbm["res"],
    "hbm_bytes_per_launch": (hbm["nores"] + hbm["res"]) / 2,
    "algorithmic_bytes_per_launch": (px * (16 * C + 4 * K) + px * (20 * C + 4 * K)) / 2,
}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "sean_fwd_pmc.json" if B == 16 else "sean_fwd_pmc_b%d.json" % B)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
