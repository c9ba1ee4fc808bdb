#!/usr/bin/env python
"""Tile-order sweep of the two dominant GEMM kernels (round 6): per NAR shape, group depth GM x {row-fastest,
column-fastest} (vx_common.h tile_walk; VX_GEMM_WALK="<gm>[,c]" is read once per process, so every point is its own process).

    python tools/gemm_walk_sweep.py time [reps]          -> us per launch + TF per (kernel, shape, walk)          (one table)
    python tools/gemm_walk_sweep.py one KERNEL            -> runs the four shapes once on KERNEL under the current VX_GEMM_WALK
                                                            (the command a `rocprofv3 --pmc FETCH_SIZE` pass wraps, see gpu_call.sh walk)
kernels: 15 = gemm_f16x2_w128_kernel (256 x 256 tiles, 4 waves), 14 = gemm_f32_dma_kernel<256, 256>
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = 31616
SHAPES = ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))
WALKS = ["default", "2", "3", "4", "8", "16", "2,c", "3,c", "4,c", "8,c"]


def one(kernel: int, reps: int):
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    for (N, K) in SHAPES:
        us, md = eng.bench_gemm(M, N, K, kernel, reps)
        print(f"RESULT {kernel} {N} {K} {us:.2f} {md:.3e}", flush=True)


def ab(rounds: int, reps: int):
    """candidates against the pre-round-6 order (-1 = 8 deep, row-fastest), interleaved in ONE process: per shape, `rounds` rounds of
    [old, cand1, cand2, ...]; prints every round and the per-candidate median ratio to `old` of the same round"""
    import statistics
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    cands = os.environ.get("WALK_CANDS", "-1 0 8,c 4,c 2,c").split()
    for kernel, name in ((15, "gemm_f16x2_w128"), (14, "gemm_f32_dma<256,256>")):
        for (N, K) in SHAPES:
            ratios = {w: [] for w in cands}
            for r in range(rounds):
                row = {}
                for w in cands:
                    os.environ["VX_BENCH_GEMM_WALK"] = w
                    row[w] = eng.bench_gemm(M, N, K, kernel, reps)[0]
                for w in cands:
                    ratios[w].append(row[w] / row[cands[0]])
                print(f"{name} N={N} K={K} round {r}: " + "  ".join(f"{w}: {row[w]:7.1f}" for w in cands), flush=True)
            print(f"{name} N={N} K={K} MEDIAN ratio to '{cands[0]}': " + "  ".join(f"{w}: {statistics.median(ratios[w]):.4f}" for w in cands), flush=True)
    os.environ.pop("VX_BENCH_GEMM_WALK", None)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    if mode == "ab":
        ab(int(sys.argv[2]) if len(sys.argv) > 2 else 5, int(sys.argv[3]) if len(sys.argv) > 3 else 6)
        return
    if mode == "one":
        one(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3)
        return
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print(f"M = {M}; us per launch (TF); walk = group depth of row tiles[,c = column-fastest inside the group]; default = GM 8 row-fastest")
    for kernel, name in ((15, "gemm_f16x2_w128"), (14, "gemm_f32_dma<256,256>")):
        print(f"== {name}")
        print("walk     " + "".join(f"  N={n:<5d}K={k:<5d}      " for n, k in SHAPES))
        for w in WALKS:
            env = dict(os.environ)
            env.pop("VX_GEMM_WALK", None)
            if w != "default":
                env["VX_GEMM_WALK"] = w
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(kernel), str(reps)], env=env, capture_output=True, text=True)
            cells = []
            for line in r.stdout.splitlines():
                if line.startswith("RESULT"):
                    _, _, n, k, us, md = line.split()
                    cells.append(f"{float(us):8.1f} ({2.0 * M * int(n) * int(k) / float(us) / 1e6:6.1f}) d{float(md):.0e}")
            print(f"{w:8s} " + "  ".join(cells) + ("" if r.returncode == 0 else f"   rc {r.returncode}: {r.stderr[-200:]}"), flush=True)


if __name__ == "__main__":
    main()
