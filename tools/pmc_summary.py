#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes per kernel.

    python tools/pmc_summary.py OUT.json NAME=DIR [NAME=DIR ...]

Each DIR is the `-d` directory of one `rocprofv3 --pmc <COUNTER> -- python bench.py ...` pass (one counter
per pass, as MI355X_MICROARCH.md prescribes).  For every kernel the average counter value per launch and the
average launch duration are written; FETCH_SIZE / WRITE_SIZE are reported in KB as the counter delivers them
(the x2 gfx950 correction of FETCH_SIZE is applied by the reader, bench.py:pmc_traffic)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(attn_fwd_w4_kernel<[^>]*>|gemm_pk_kernel<[^>]*>|gemm_w4_kernel<[^>]*>|attn_fwd_v2_kernel<[^>]*>|attn_fwd_kernel<[^>]*>|gemm256_kernel<[^>]*>|gemm_bf16_kernel<[^>]*>|"
                  r"ln_modulate_rows_kernel<[^>]*>|ln_modulate_kernel|rmsnorm_rope_rows_kernel|rmsnorm_rope_kernel|rmsnorm_rope|conv_cl_kernel<[^>]*>|conv3_patch_kernel|conv3_head_kernel<[^>]*>|attn_combine_kernel)", name)
    return m.group(1) if m else name[:60]


def main():
    out, passes = sys.argv[1], dict(a.split("=", 1) for a in sys.argv[2:])
    res = defaultdict(dict)
    for pname, d in passes.items():
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc = defaultdict(lambda: [0, 0.0, 0.0, 0])
        for f in files:
            for row in csv.DictReader(open(f)):
                k = (short(row["Kernel_Name"]), row.get("Counter_Name", pname))
                a = acc[k]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                a[3] = int(row["Grid_Size"])
        names = {c for (_, c) in acc}
        for (k, c), (n, v, ms, grid) in acc.items():
            # one counter per pass keeps the pass name (fetch / write); several counters are keyed by their own names
            key = pname if len(names) == 1 else c
            res[k][key] = {"launches_counted": n, "avg_counter": v / n, "avg_ms": ms / n, "grid": grid}
    keep = {k: v for k, v in res.items() if any(p["avg_ms"] > 0.05 for p in v.values())}
    # SQ passes: sustained clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs), matrix-pipe busy fraction (busy cycles summed over the
    # 1024 SIMDs against the per-XCD active cycles), share of wave cycles spent waiting
    derived = {}
    for k, v in keep.items():
        if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            act = v["GRBM_GUI_ACTIVE"]["avg_counter"] / 8.0
            d = {"avg_ms": round(v["GRBM_GUI_ACTIVE"]["avg_ms"], 3), "clock_GHz": round(act / (v["GRBM_GUI_ACTIVE"]["avg_ms"] * 1e-3) / 1e9, 3),
                 "mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_counter"] / (act * 1024.0), 3)}
            if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
                d["wait_any_frac_of_wave_cycles"] = round(v["SQ_WAIT_ANY"]["avg_counter"] / v["SQ_WAVE_CYCLES"]["avg_counter"], 3)
            derived[k] = d
    if derived:
        keep["_derived"] = derived
    json.dump(keep, open(out, "w"), indent=1)
    keep.pop("_derived", None)
    for k, v in sorted(keep.items(), key=lambda kv: -max(p["avg_ms"] * p["launches_counted"] for p in kv[1].values()))[:12]:
        print(k, {p: (round(x["avg_counter"]), round(x["avg_ms"], 3)) for p, x in v.items()})


if __name__ == "__main__":
    main()
