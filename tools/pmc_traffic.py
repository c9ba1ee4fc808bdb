#!/usr/bin/env python
"""HBM-side read traffic per launch of the hot kernels, from a rocprofv3 PMC pass of bench.py -> profiles/rNN_pmc_<kernel>.json
(the files bench.py's `roofline.traffic` quotes).  Recipe (on an MI355X, counters in their own pass, under `timeout`):

    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_pmc -o pmc -- \
        python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile
    python $REPO/tools/rocpd_pmc_summary.py $(find $REPO/gpurun_out/prof_pmc -name '*.db' | head -1) > $REPO/gpurun_out/pmc_fetch.csv
    python $REPO/tools/pmc_traffic.py $REPO/gpurun_out/pmc_fetch.csv 02          # writes profiles/r02_pmc_*.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the 128-B requests of wide coalesced streaming reads at
64 B -> x2 for these kernels (all of them read with 16 B per lane).  The counter unit is KB.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = {  # json suffix -> (kernel-name substring, what the algorithmic bytes of one launch are)
    "dec_attn": "dec_attn_kernel",
    "gemm_f16x2": "gemm_f16x2_",            # gemm_f16x2_w128_kernel (long row sets) + gemm_f16x2_kernel<...> (short ones), launch-weighted
    "skinny_gemm": "skinny_gemm_kernel",
    "skinny16": "skinny16_relu_pack_kernel",
    "attn_full_x3": "attn_full_x3_kernel",
    "attn_full_h2": "attn_full_h2_kernel",
    "gemm_f32": "gemm_f32_",                # the reference-arithmetic leg (bench.py --arith f32): every projection -- the LDS-DMA
                                            # kernels gemm_f32_dma_kernel<256 / 128> and the register-staged gemm_f32_kernel, launch-weighted
    "attn_full": "attn_full_kernel",        # and the exact-fp32 full-sequence attention
    "gemm_bf16x3": "gemm_bf16x3_dma_kernel",
}


def algo_dec_attn():
    """run average of the algorithmic bytes of one dec_attn launch in the bench workload: sum_b 8192 B x ctx_b (K and V rows of
    16 heads x 64 x 4 B), ctx_b = S_b + 1 + Tp_b + t averaged over the 600 steps, + W_o (4 MiB) for the fused out_proj."""
    import bench
    rows = bench.make_rows(0, bench.ROWS_PER_GPU)
    tot = 0.0
    for r in rows:
        L = len(r["text"]) + 1 + r["prompt"].shape[0]
        tot += sum(8192.0 * (L + t) for t in range(1, bench.FRAMES + 1)) / bench.FRAMES
    return tot + 1024 * 1024 * 4


def main(csv_path, rnd, arith=""):
    """arith: the --arith the profiled bench command ran with ("" = the product default); it is part of the `source` string and
    keeps a Vocos-only gemm_f32 average of a default-arithmetic pass from being mistaken for the fp32 leg's."""
    import bench
    rows = list(csv.DictReader(open(csv_path)))
    for key, sub in KERNELS.items():
        if key in ("gemm_f32", "attn_full") and arith != "f32":
            continue
        if key == "gemm_bf16x3" and arith != "bf16x3":
            continue
        sel = [r for r in rows if sub in r["kernel"] and r["counter"] == "FETCH_SIZE"]
        if not sel:
            continue
        n = sum(int(r["dispatches"]) for r in sel)
        avg_kb = sum(float(r["avg"]) * int(r["dispatches"]) for r in sel) / n
        # one entry per kernel SYMBOL of the class (round 5 quoted the 8-wave kernel's figure for the 4-wave kernel: the class average
        # and the prose were read off different rows of the same CSV -- the rows are now part of the file the prose is generated from)
        by_symbol = [{"kernel": r["kernel"], "launches": int(r["dispatches"]), "traffic_bytes_per_launch": int(float(r["avg"]) * 1024 * 2),
                      "avg_dispatch_us": float(r["avg_dispatch_us"])} for r in sel]
        out = {"kernel": sub, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 0 "
                                        "--no-cpu-baseline --no-profile --no-ref-arith" + (f" --arith {arith}" if arith else "")
                                        + " (tools/pmc_traffic.py)",
               "launches": n, "avg_FETCH_SIZE_KB": round(avg_kb, 1),
               "correction": "x2: gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) streaming reads at 64 B "
                             "(MI355X_MICROARCH.md, HBM)",
               "traffic_bytes_per_launch": int(avg_kb * 1024 * 2), "by_symbol": by_symbol,
               # bench.py refuses the file once the kernel's translation unit (or vx_common.h) has changed
               "source_sha256": bench.kernel_source_digest(key)}
        if key == "dec_attn":
            out["algo_bytes_per_launch_same_run"] = int(algo_dec_attn())
            out["note"] = "launches of inactive steps (after the forced EOS) are included in the average with ~0 bytes"
        p = os.path.join(ROOT, "profiles", f"r{rnd}_pmc_{key}.json")
        json.dump(out, open(p, "w"), indent=1)
        print(p, out["traffic_bytes_per_launch"])


def table(csv_path):
    """markdown table, one row per kernel symbol: launches, FETCH_SIZE per launch (x2 corrected), launch time under the counter pass"""
    rows = [r for r in csv.DictReader(open(csv_path)) if r["counter"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: -float(r["avg"]) * int(r["dispatches"]))
    out = ["| kernel symbol | launches | FETCH per launch (MB, x2 corrected) | total (GB) | avg launch under the counter pass (us) |", "|---|---|---|---|---|"]
    for r in rows:
        mb = float(r["avg"]) * 1024 * 2 / 1e6
        out.append(f"| `{r['kernel']}` | {r['dispatches']} | {mb:.1f} | {mb * int(r['dispatches']) / 1e3:.2f} | {r['avg_dispatch_us']} |")
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--table":
        print(table(sys.argv[2]))
        sys.exit(0)
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "02", sys.argv[3] if len(sys.argv) > 3 else "")
