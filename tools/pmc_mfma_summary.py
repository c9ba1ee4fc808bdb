#!/usr/bin/env python
"""MFMA / VALU busy fractions of the matrix kernels from the three rocprofv3 --pmc passes of tools/gpu_call.sh mfma
(gpurun_out/pm_{1,2,3}.csv) -> JSON on stdout (kept as profiles/rNN_mfma_busy.json).

Units (MI355X_MICROARCH.md, rocprofv3 PMC section): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles, summed over all SIMDs (exactly
32 per v_mfma_f32_32x32x16_f16: the file shows BUSY / SQ_INSTS_MFMA = 32.0); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles; GRBM_GUI_ACTIVE counts cycles per XCD and is summed over the 8 XCDs.  1024 SIMDs = 256 CUs x 4.
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)       valu_busy = 4 * SQ_ACTIVE_INST_VALU / (same)"""
import csv
import json
import sys

# only the long (>= 200 us) kernels: GRBM_GUI_ACTIVE of a 7-36 us decode kernel includes the idle ramp around it (implied clocks of
# 2.8-4.6 GHz), so a busy fraction relative to it means nothing there
KERNELS = {"gemm_f16x2 (256x256 tiles, 4 waves of 128x128)": "gemm_f16x2_w128_kernel", "gemm_f16x2 (256x256 tiles, 8 waves of 64x128)": "gemm_f16x2_kernel<256, 256", "attn_full_h2": "attn_full_h2_kernel",
           "gemm_f32 register-staged (Vocos head, short row sets)": "gemm_f32_kernel",
           "gemm_f32 LDS-DMA 256 x 256 (every long projection with --arith f32)": "gemm_f32_dma_kernel<256, 256>",
           "gemm_f32 LDS-DMA 256 x 128 (long row sets, N % 256 != 0: Vocos)": "gemm_f32_dma_kernel<256, 128>",
           "gemm_f32 LDS-DMA 128 x 128 (--arith f32, prefill)": "gemm_f32_dma_kernel<128, 128>", "attn_full (--arith f32)": "attn_full_kernel",
           "gemm_bf16x3_dma (--arith bf16x3)": "gemm_bf16x3_dma_kernel", "attn_full_x3 (--arith bf16x3)": "attn_full_x3_kernel"}


def main(paths):
    rows = {}
    for p in paths:
        for r in csv.DictReader(open(p)):
            rows.setdefault(r["kernel"], {})[r["counter"]] = (float(r["avg"]), float(r["avg_dispatch_us"]), int(r["dispatches"]))
    out = {"source": "three separate `rocprofv3 --kernel-trace --pmc ...` passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                     "--no-profile` (tools/gpu_call.sh mfma); per-launch averages over the launches of the batch", "kernels": {}}
    for label, sub in KERNELS.items():
        k = next((n for n in rows if sub in n), None)
        if k is None or "GRBM_GUI_ACTIVE" not in rows[k]:
            continue
        d = rows[k]
        cyc = d["GRBM_GUI_ACTIVE"][0] / 8.0
        e = {"launches": d["GRBM_GUI_ACTIVE"][2], "avg_us_under_profiler": d["GRBM_GUI_ACTIVE"][1], "gpu_cycles_per_launch": round(cyc),
             "implied_clock_mhz": round(cyc / d["GRBM_GUI_ACTIVE"][1]),
             "mfma_busy": round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0,))[0] / (1024 * cyc), 4)}
        if "SQ_ACTIVE_INST_VALU" in d:
            e["valu_busy"] = round(4 * d["SQ_ACTIVE_INST_VALU"][0] / (1024 * cyc), 4)
        if "SQ_INSTS_MFMA" in d and d["SQ_INSTS_MFMA"][0] > 0:
            e["mfma_instructions_per_launch"] = round(d["SQ_INSTS_MFMA"][0])
            e["busy_cycles_per_mfma"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / d["SQ_INSTS_MFMA"][0], 2)
        if "SQ_WAVE_CYCLES" in d:
            e["wave_time_issue_stalled"] = round(d["SQ_WAIT_INST_ANY"][0] / d["SQ_WAVE_CYCLES"][0], 4)
        out["kernels"][label] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:] or [f"gpurun_out/pm_{i}.csv" for i in (1, 2, 3)])
