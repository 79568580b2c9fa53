"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family -> JSON (run on the GPU box after the PMC passes).
usage: python tools/pmc_summarize.py <dir with *counter_collection.csv> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:vtp::)?(\w+)", name.replace("_ZN3vtp", ""))
    if "gemm8p_grouped_tn_kernel" in name:
        return "gemm8p_grouped_tn"
    if "gemm8p_kernel" in name:
        return "gemm8p_tn" if re.search(r"gemm8p_kernel<\d+, true", name) else "gemm8p_nt"
    if "gemm_nt_kernel" in name:
        return "gemm_tn" if re.search(r"gemm_nt_kernel<[^>]*, true, (true|false)>", name) else "gemm_nt"
    for k in ("attn_fwd", "attn_bwd_fused", "attn_bwd_dq", "attn_bwd_dkv", "norm_bwd", "norm_fwd", "adamw", "prep_weights", "colsum_bf16",
              "reduce_slabs", "swiglu_bwd", "rope_qk", "softmax_center", "dino_ce", "ema_kernel"):
        if k in name:
            return k
    return m.group(1) if m else name[:40]


def main(src, out):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                fam = family(row["Kernel_Name"])
                c = row["Counter_Name"]
                acc[fam][c] += float(row["Counter_Value"])
                calls[fam][c] += 1
    res = {}
    for fam in acc:
        res[fam] = {c: {"sum": acc[fam][c], "dispatches": calls[fam][c], "per_dispatch": acc[fam][c] / max(1, calls[fam][c])}
                    for c in acc[fam]}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for fam in sorted(res, key=lambda k: -sum(v["sum"] for v in res[k].values()))[:12]:
        print(fam, {c: round(v["per_dispatch"], 1) for c, v in res[fam].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
