"""CPU (cross-compiles, no GPU): properties of the generated gfx950 code that the design counts on and that a source edit or a
compiler update can lose silently.
  * dec_attn (the dominant kernel, 7 200 launches per batch) reaches its first memory requests without a dependent round trip: every
    operand of those requests lies in the 16 preloaded kernarg dwords (vall-e-x_amd/_build.py), the slot record is requested first and
    the first K / V tile right behind it, and the first wait lets the tile stay in flight;
  * the decode GEMMs request a whole round of x and (non-temporal) weight blocks before their first MFMA -- the step is a chain of memory
    round trips, a load sunk to its use is one more of them;
  * the steady-state loop of the 256 x 256 f16x2 GEMM is what section 5 describes per pair of K tiles: 2 rendezvous, 2 x 48 MFMAs
    (3 per 32 x 32 x 16 block), 2 x 24 fragment reads, 2 x 8 LDS-DMA requests spread between the MFMAs, nothing spilled;
  * the tile loop of the f16x2 attention: per pair of key tiles 2 barriers, 2 x 24 MFMAs, 2 x 16 fragment reads, one v_exp per score."""
import importlib.util
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _product_flags(src):
    spec = importlib.util.spec_from_file_location("_vx_build", os.path.join(ROOT, "vall-e-x_amd", "_build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m._hipcc(), m.FLAGS + m.EXTRA_FLAGS.get(src, [])


_ASM = {}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    """source name -> lines of the device assembly the PRODUCT flags give it (compiled once per module)"""
    def get(src):
        if src not in _ASM:
            hipcc, flags = _product_flags(src)
            if shutil.which(hipcc) is None and not os.path.exists(hipcc):
                pytest.skip("no hipcc")
            out = tmp_path_factory.mktemp("isa") / (src + ".s")
            r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", str(out),
                                                  os.path.join(ROOT, "vall-e-x_amd", "csrc", src)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            _ASM[src] = out.read_text().splitlines()
        return _ASM[src]
    return get


def kernel_body(lines, symbol_re):
    """instructions and labels of one kernel (comments and directives stripped)"""
    start = next(i for i, ln in enumerate(lines) if re.match(symbol_re + r"\w*:", ln))
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith("\ts_endpgm") or ln.startswith(".Lfunc_end"):
            break
        ln = ln.split(";")[0].rstrip()
        if ln.strip():
            body.append(ln.strip())
    return body


def loops(body):
    """every [label .. backward branch to it] span of a kernel body"""
    pos = {ln[:-1]: i for i, ln in enumerate(body) if ln.endswith(":")}
    out = []
    for i, ln in enumerate(body):
        m = re.match(r"s_c?branch\w*\s+(\.\w+)", ln)
        if m and pos.get(m.group(1), i) < i:
            out.append(body[pos[m.group(1)]: i + 1])
    return out


def count(seq, pattern):
    return sum(1 for ln in seq if re.match(pattern, ln))


def metadata(lines, name_substr, key):
    """a field of the kernel's record in the amdhsa.kernels metadata (the compiler's own figure); a record starts at `  - .` and its
    keys are sorted, so some of them precede `.name`"""
    at = next(i for i, ln in enumerate(lines) if ".name:" in ln and name_substr in ln)
    lo = at
    while not lines[lo].lstrip().startswith("- ."):
        lo -= 1
    hi = at + 1
    while hi < len(lines) and not lines[hi].lstrip().startswith("- .") and not lines[hi].startswith("amdhsa."):
        hi += 1
    for ln in lines[lo:hi]:
        t = ln.strip().lstrip("- ")
        if t.startswith("." + key + ":"):
            return int(t.split(":")[1])
    raise KeyError((name_substr, key))


def test_fused_dec_attn_reaches_its_first_requests_without_a_round_trip(asm):
    body = kernel_body(asm("decode.hip"), r"_ZN2vx15dec_attn_kernelILb1ELi4ELb0EEE")
    # with kernarg preload the hardware enters 256 bytes behind the symbol: skip the compatibility header (s_load ... s_branch)
    entry = next(i for i, ln in enumerate(body) if ln.startswith("s_branch")) + 1
    head = body[entry:]
    first_wait = next(i for i, ln in enumerate(head) if ln.startswith("s_waitcnt") and "vmcnt" in ln)
    before = head[:first_wait]
    assert not [ln for ln in before if ln.startswith("s_load")], "an argument of the head-of-kernel requests is not preloaded"
    loads = [ln for ln in before if ln.startswith("global_load_dwordx4")]
    assert len(loads) == 9 and " nt" not in loads[0] and all(" nt" in ln for ln in loads[1:]), loads     # record, then 4 K + 4 V rows
    assert re.search(r"vmcnt\(8\)", head[first_wait]), head[first_wait]                                   # waits for the record only


def test_decode_gemms_request_a_whole_round_before_their_first_mfma(asm):
    lines = asm("decode.hip")
    for sym, name, n_loads, n_nt in ((r"_ZN2vx18skinny_gemm_kernelE", "skinny_gemm_kernel", 16, 8),             # 8 x blocks + 8 weight blocks
                                     (r"_ZN2vx25skinny16_relu_pack_kernelE", "skinny16_relu_pack_kernel", 8, 4)):
        body = kernel_body(lines, sym)
        first_mfma = next(i for i, ln in enumerate(body) if ln.startswith("v_mfma"))
        loads = [ln for ln in body[:first_mfma] if ln.startswith("global_load_dwordx4")]
        assert len(loads) >= n_loads, (name, len(loads))
        assert sum(" nt" in ln for ln in loads) >= n_nt, (name, "the weight stream must be non-temporal")
        assert metadata(lines, name, "private_segment_fixed_size") == 0, name
    # the streaming loop of dec_attn keeps a tile of each of K and V in flight per iteration, all of it non-temporal
    stream = max(loops(kernel_body(lines, r"_ZN2vx15dec_attn_kernelILb1ELi4ELb0EEE")), key=len)
    ld = [ln for ln in stream if ln.startswith("global_load_dwordx4")]
    assert len(ld) == 16 and all(" nt" in ln for ln in ld), ld
    assert count(stream, r"s_barrier") == 0 and count(stream, r"ds_") == 0          # q . k through DPP row sums: no LDS, no barrier


def test_f16x2_gemm_steady_state_loop(asm):
    lines = asm("gemm_f16x2.hip")
    body = kernel_body(lines, r"_ZN2vx17gemm_f16x2_kernelILi256ELi256ELi2ELi0EEE")
    main = max(loops(body), key=lambda lp: count(lp, r"v_mfma"))
    # one iteration = two K tiles of 32 (the two LDS stages trade places)
    assert count(main, r"v_mfma_f32_32x32x16_f16") == 96          # 8 blocks x 3 products x 2 k16 steps x 2 tiles per wave
    assert count(main, r"s_barrier") == 2                          # ONE rendezvous per K tile
    assert count(main, r"ds_read_b128") == 48                      # (2 + 4) operand blocks x 2 planes x 2 k16 steps x 2 tiles
    assert count(main, r"global_load_lds_dwordx4") == 16           # 8 KiB per wave and stage, one request behind every sixth MFMA
    assert count(main, r"scratch_|buffer_store|buffer_load") == 0
    # the requests are spread: never two LDS-DMA instructions without an MFMA between them
    kinds = [("d" if ln.startswith("global_load_lds") else "m") for ln in main if ln.startswith(("global_load_lds", "v_mfma"))]
    assert "dd" not in "".join(kinds)
    assert metadata(lines, "gemm_f16x2_kernelILi256ELi256ELi2ELi0E", "private_segment_fixed_size") == 0
    assert metadata(lines, "gemm_f16x2_kernelILi256ELi256ELi2ELi0E", "vgpr_count") <= 256          # two waves per SIMD
    assert metadata(lines, "gemm_f16x2_kernelILi256ELi256ELi2ELi0E", "group_segment_fixed_size") == 131072


def test_f16x2_attention_tile_loop(asm):
    lines = asm("attn_full_h2.hip")
    body = kernel_body(lines, r"_ZN2vx19attn_full_h2_kernelILi0EEE")
    main = max(loops(body), key=lambda lp: count(lp, r"v_mfma"))
    # one iteration = two key tiles of 32 (score registers and LDS buffers alternate roles)
    assert count(main, r"v_mfma_f32_32x32x16_f16") == 48          # per tile: 12 for S^T (one chain) + 12 for O^T (two chains)
    assert count(main, r"s_barrier") == 2                          # one barrier per tile
    assert count(main, r"ds_read_b128") == 32                      # per tile: 4 k-steps x 2 K planes + 2 k-steps x 2 halves x 2 V planes
    assert count(main, r"v_exp_f32") == 34                         # 16 scores per lane and tile + the rescale of the running maximum
    assert count(main, r"global_load_dwordx4") == 8                # the staged K / V rows of tile t + 3 (two of each per thread)
    assert count(main, r"scratch_") == 0
    assert metadata(lines, "attn_full_h2_kernelILi0E", "private_segment_fixed_size") == 0
    assert metadata(lines, "attn_full_h2_kernelILi0E", "vgpr_count") <= 256                          # two 4-wave workgroups per CU = two waves per SIMD


def test_fp32_dma_gemm_tile_loop(asm):
    """the LDS-DMA fp32 GEMM of the reference-arithmetic mode (round 5): per K tile of 32 ONE rendezvous (counted wait + bare barrier),
    64 (128 for the 256 x 256 tile) fp32 MFMAs per wave (2 x 2 or 2 x 4 blocks x 16 k2 steps), 16 (24) fragment reads, the stage's LDS-DMA
    requests (6 x 1 KiB per wave for the 256 x 128 tile), no ds_write pass, nothing spilled, two waves per SIMD"""
    lines = asm("gemm_f32.hip")
    for sym, name, ndma, lds, nj in ((r"_ZN2vx19gemm_f32_dma_kernelILi256ELi128EEE", "gemm_f32_dma_kernelILi256ELi128E", 6, 98304, 2),
                                     (r"_ZN2vx19gemm_f32_dma_kernelILi128ELi128EEE", "gemm_f32_dma_kernelILi128ELi128E", 8, 65536, 2),
                                     (r"_ZN2vx19gemm_f32_dma_kernelILi256ELi256EEE", "gemm_f32_dma_kernelILi256ELi256E", 8, 131072, 4)):
        body = kernel_body(lines, sym)
        main = max(loops(body), key=lambda lp: count(lp, r"v_mfma"))
        assert count(main, r"v_mfma_f32_32x32x2_f32") == 32 * nj, name       # 2 x nj blocks x 16 k2 steps
        assert count(main, r"s_barrier") == 1, name
        assert count(main, r"ds_read_b128") == 4 * (2 + nj) and count(main, r"ds_write") == 0, name
        # the compiler lays the `kt + 1 < nk` request block out behind the loop's backward branch: count over the kernel = prologue + loop
        assert count(body, r"global_load_lds_dwordx4") == 2 * ndma, name
        assert count(main, r"scratch_|buffer_store|buffer_load") == 0, name
        # the fragments of a k-block are waited for with the NEXT block's four reads still in flight, never with vmcnt(0) in between
        waits = [ln for ln in main if ln.startswith("s_waitcnt")]
        assert sum("vmcnt(0)" in w for w in waits) == 1 and sum(f"lgkmcnt({2 + nj})" in w for w in waits) >= 3, waits
        assert metadata(lines, name, "private_segment_fixed_size") == 0
        assert metadata(lines, name, "vgpr_count") <= (128 if nj == 2 else 256) and metadata(lines, name, "group_segment_fixed_size") == lds


def test_f16x2_four_wave_gemm_steady_state_loop(asm):
    """round 5: the 256 x 256 tile on four waves of 128 x 128 (the product's kernel for the long row sets): per K tile of 32 ONE
    rendezvous between the two k16 steps, 96 MFMAs per wave (16 blocks x 3 products x 2 steps) on accumulators that stay in AGPRs (no
    v_accvgpr move in the loop), 32 fragment reads (0.33 per MFMA) and the 16 LDS-DMA requests of tile kt + 2, each of them behind an
    MFMA -- never two memory instructions back to back --, nothing spilled, one wave per SIMD"""
    lines = asm("gemm_f16x2.hip")
    body = kernel_body(lines, r"_ZN2vx22gemm_f16x2_w128_kernelE")
    main = max(loops(body), key=lambda lp: count(lp, r"v_mfma"))
    assert count(main, r"v_mfma_f32_32x32x16_f16") == 96
    assert count(main, r"s_barrier") == 1
    assert count(main, r"ds_read_b128") == 32 and count(main, r"global_load_lds_dwordx4") == 16
    assert count(main, r"v_accvgpr") == 0 and count(body, r"scratch_") == 0
    waits = [ln for ln in main if ln.startswith("s_waitcnt")]
    assert len(waits) == 2 and sum("vmcnt(0)" in w for w in waits) == 1, waits          # the set of this step; the rendezvous
    kinds = "".join("m" if ln.startswith("v_mfma") else "x" for ln in main if ln.startswith(("v_mfma", "ds_read_b128", "global_load_lds")))
    assert "xxx" not in kinds                                      # at most a read + a request between two MFMAs
    assert metadata(lines, "gemm_f16x2_w128_kernel", "private_segment_fixed_size") == 0
    assert metadata(lines, "gemm_f16x2_w128_kernel", "vgpr_count") <= 512 and metadata(lines, "gemm_f16x2_w128_kernel", "group_segment_fixed_size") == 131072
