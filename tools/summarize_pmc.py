#!/usr/bin/env python3
"""Aggregate the rocprofv3 PMC passes of bench.py (and of tools/pmc_conv_shapes.py) into profiles/rNN_traffic.json.
usage: summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel_stats.csv> <out.json>
                        [<shapes fetch csv> <shapes write csv> <pmc_conv_shapes.json>]
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md).
`per_family` is keyed like bench.py's roofline.families (one entry = one ops.* call: GroupNorm = its stats + apply
launches together, linear / conv3x3 include their split-K reduce launches); `per_kernel` keeps the finer split."""
import collections, csv, json, sys


def kernel_class(k):
    if "sdmi_gemm_kernel<1" in k or "conv3x3_narrow_kernel" in k or "sdmi_gemm_wide_kernel<1" in k or "sdmi_conv_halo_kernel" in k: return "conv"
    if "sdmi_gemm_kernel<0" in k or "sdmi_gemm_wide_kernel<0" in k: return "linear_tiled"
    if "sdmi_gemm_panel_kernel" in k: return "linear_panel"
    if "splitk_reduce" in k: return "splitk_reduce"
    if "attn_fwd" in k: return "attention"
    if "vae_attn" in k: return "attention_vae"
    if "gn_apply" in k: return "gn_apply"
    if "gn_stats" in k and "true>" in k: return "gn_fused"
    if "gn_stats" in k: return "gn_stats"
    if "layernorm" in k: return "layernorm"
    return None


FAMILY = {"conv3x3": ("conv",), "linear": ("linear_tiled", "linear_panel"), "attention": ("attention",),
          "groupnorm": ("gn_stats", "gn_fused", "gn_apply"), "layernorm": ("layernorm",)}
CALLS = {"conv3x3": ("conv",), "linear": ("linear_tiled", "linear_panel"), "attention": ("attention",),
         "groupnorm": ("gn_stats", "gn_fused"), "layernorm": ("layernorm",)}     # kernels that occur once per ops.* call


def rows(path, name):
    rs = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rs


def agg(path, name):
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in rows(path, name):
        f = kernel_class(r["Kernel_Name"])
        if f:
            out[f][0] += 1
            out[f][1] += float(r["Counter_Value"])
    return out


fetch, write = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[3])):
    f = kernel_class(r["Name"])
    if f:
        dur[f][0] += int(r["Calls"])
        dur[f][1] += float(r["TotalDurationNs"])
res = {"note": "L2-miss bytes per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
               "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-secondary --no-legs` (round 3: hoisted time path, dup_rows prefix, GEGLU on the 256 x 256 x 32 tile at K <= 640, GroupNorm butterfly reduction); FETCH_SIZE doubled "
               "per MI355X_MICROARCH.md; Infinity-Cache hits included (an upper bound on HBM bytes)",
       "unit_note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); average "
                    "durations from the --kernel-trace --stats run of the same build",
       "per_family": {}, "per_kernel": {}}
for f in fetch:
    n, fs = fetch[f]
    ws = write[f][1]
    b = (2 * fs + ws) * 1024 / n
    us = dur[f][1] / dur[f][0] / 1e3 if dur[f][0] else None
    res["per_kernel"][f] = {"launches": n, "fetch_kib_raw": fs / n, "write_kib": ws / n, "bytes_per_launch": int(b),
                            "avg_launch_us": us, "l2_miss_TB_per_s": b / us / 1e6 if us else None}
for fam, ks in FAMILY.items():
    calls = sum(fetch[k][0] for k in CALLS[fam] if k in fetch)
    if not calls:
        continue
    extra = ("splitk_reduce",) if fam == "conv3x3" else ()       # the reduce launches in the step are the convs' (K >= 11520)
    tot = sum((2 * fetch[k][1] + write[k][1]) * 1024 for k in ks + extra if k in fetch)
    t_ns = sum(dur[k][1] for k in ks + extra) / max(1, sum(dur[k][0] for k in CALLS[fam])) if dur else None
    res["per_family"][fam] = {"calls": calls, "bytes_per_launch": int(tot / calls),
                              "avg_call_us": t_ns / 1e3 if t_ns else None,
                              "l2_miss_TB_per_s": tot / calls / t_ns / 1e3 if t_ns else None}

if len(sys.argv) > 7:
    table = json.load(open(sys.argv[7]))
    fr, wr = rows(sys.argv[5], "FETCH_SIZE"), rows(sys.argv[6], "WRITE_SIZE")

    def per_shape(rs):
        acc, idx, seen = [0.0] * len(table), 0, 0
        for r in rs:
            c = kernel_class(r["Kernel_Name"])
            if c == "conv":
                if seen == table[idx]["launches"]:
                    idx, seen = idx + 1, 0
                seen += 1
                acc[idx] += float(r["Counter_Value"])
            elif c == "splitk_reduce":
                acc[idx] += float(r["Counter_Value"])
        return acc
    fa, wa = per_shape(fr), per_shape(wr)
    res["conv_shapes"] = []
    for e, f_, w_ in zip(table, fa, wa):
        b = (2 * f_ + w_) * 1024 / e["launches"]
        res["conv_shapes"].append({"shape_b_h_w_cin_cout": e["shape"], "M": e["M"], "N": e["N"], "K": e["K"],
                                   "algorithmic_bytes": e["algorithmic_bytes"], "l2_miss_bytes_per_launch": int(b),
                                   "fetch_bytes": int(2 * f_ * 1024 / e["launches"]), "write_bytes": int(w_ * 1024 / e["launches"]),
                                   "ratio": b / e["algorithmic_bytes"]})
json.dump(res, open(sys.argv[4], "w"), indent=1)
for sec in ("per_family", "per_kernel"):
    for f, v in res[sec].items():
        print(sec, f, v)
for v in res.get("conv_shapes", []):
    print(v)
