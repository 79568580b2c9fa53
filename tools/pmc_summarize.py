"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family -> JSON (run on the GPU box after the PMC passes).
usage: python tools/pmc_summarize.py <dir with *counter_collection.csv | summary.json> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:vtp::)?(\w+)", name.replace("_ZN3vtp", ""))
    if "gemm4w_grouped_tn" in name:  # (..._kernel: uniform tiles x slices; ..._items_kernel: the work-item list of round 6)
        return "gemm4w_grouped_tn"
    if "gemm4w_kernel" in name:
        return "gemm4w_nt"
    if "gemm8p_grouped_tn_kernel" in name:
        return "gemm8p_grouped_tn"
    if "gemm8h_kernel" in name:
        return "gemm8h_nt"
    if "gemm8p_dyn_kernel" in name:  # the persistent NT launch with tiles drawn from queues (round 6)
        return "gemm8p_nt"
    if "gemm8p_kernel" in name:
        return "gemm8p_tn" if re.search(r"gemm8p_kernel<\d+, true", name) else "gemm8p_nt"
    if "gemm_nt_kernel" in name:
        return "gemm_tn" if re.search(r"gemm_nt_kernel<[^>]*, true, (true|false)>", name) else "gemm_nt"
    for k in ("attn_fwd", "attn_bwd_fused", "attn_bwd_dq", "attn_bwd_dkv", "norm_bwd", "norm_fwd", "adamw", "prep_weights", "colsum_bf16",
              "reduce_slabs", "swiglu_bwd", "rope_qk", "softmax_center", "dino_ce", "ema_kernel"):
        if k in name:
            return k
    return m.group(1) if m else name[:40]


N_XCD, N_SIMD = 8, 1024  # MI355X: 8 XCDs, 256 CUs x 4 SIMDs


def derived(c):
    """SQ-counter ratios per kernel family (only when the pass collected them).  Normalisation calibrated on this pool with ONE
    8192 x 8192 x 4096 bf16 GEMM under the same counters (profiles/r04_pmc_sq_cal.json): SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x the number
    of 32x32x16 MFMAs exactly (summed over all SIMDs); GRBM_GUI_ACTIVE comes out summed over the 8 XCDs (counter / 8 / kernel time =
    1.57 GHz).  mfma_util = fraction of SIMD-cycles the matrix pipe was busy while the kernel ran; x clock / 2.4 GHz = fraction of
    the 2.5 PFLOP/s data-sheet peak."""
    d = {}
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
    if gui > 0:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * N_SIMD), 4)
        if "SQ_ACTIVE_INST_VALU" in c:  # quad-cycles (MI355X_MICROARCH.md): x 4 = cycles
            d["valu_busy"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (gui * N_SIMD), 4)
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_active"] = round(c["SQ_LDS_IDX_ACTIVE"] / (gui * N_SIMD / 4), 4)  # per CU: one LDS per 4 SIMDs
    if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0 and "SQ_LDS_BANK_CONFLICT" in c:
        d["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    return d


def rederive(src, out):
    """add the derived ratios to an existing summary JSON (one written before they existed, or merged from several passes)"""
    res = json.load(open(src))
    for fam, d in res.items():
        d.update(derived({c: v["sum"] for c, v in d.items() if isinstance(v, dict)}))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for fam in ("gemm8p_nt", "gemm8p_tn", "gemm8p_grouped_tn", "gemm4w_grouped_tn", "gemm4w_nt", "gemm8h_nt", "gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd_fused", "norm_bwd", "adamw"):
        if fam in res:
            print(fam, {k: v for k, v in res[fam].items() if not isinstance(v, dict)})


def main(src, out):
    if src.endswith(".json"):
        return rederive(src, out)
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
        res[fam].update(derived(acc[fam]))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for fam in sorted(res, key=lambda k: -sum(v["sum"] for v in res[k].values() if isinstance(v, dict)))[:14]:
        print(fam, {c: (round(v["per_dispatch"], 1) if isinstance(v, dict) else v) for c, v in res[fam].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
