#!/usr/bin/env python
"""One command for a user who HAS the real files (the build container and the GPU boxes have no network, so every committed
test runs on seeded synthetic weights of the checkpoint's exact layout -- this is the tool that closes that gap on site):

    python tools/verify_checkpoint.py --ckpt checkpoints/vallex-checkpoint.pt \
        [--vocos path/to/pytorch_model.bin] [--reference /path/to/VALL-E-X] [--presets /path/to/VALL-E-X/presets] \
        [--max-presets 6] [--frames 150] [--ariths f16x2,f32] [--out report.json]

What it reports (JSON; exit status 0 only if every parity check that could run passed):
  load            torch.load(ckpt)["model"] (utils/generation.py:79-83) -> VALLE.load_state_dict(strict=True): key count, layers,
                  dtypes -- the reference's weight wire format taken as is;
  headroom        max |operand| the f16x2 kernels will see (LayerNorm outputs, q/8, k, v, attention output, ReLU'd FFN activations)
                  on a 12-layer prefill + one NAR stage of the first preset, measured on the CPU oracle, next to the fp16 limit 2047
                  (DESIGN.md section 3); > 2047 is NOT an error (the engine re-runs such phases in fp32), it predicts fallbacks;
  parity[arith]   per preset (prompt from the .npz, synthetic phoneme ids as text, greedy AND top-k = 10 with injected uniforms,
                  EOS forced at --frames): ids of the MI355X engine vs the REFERENCE's own VALLE.inference run on the CPU from
                  --reference (imported, its multinomial replaced by the same inverse-CDF draws) -- or, without --reference, vs the
                  engine in fp32 arithmetic; plus vx_last_fallbacks of every call;
  vocos           with --vocos: waveform RMS of the GPU head against the CPU restatement (oracle/vallex_oracle.VocosOracle) and,
                  when the pip package `vocos` is importable and --vocos-config is given, against the package itself (the pin the
                  offline build cannot have).
--no-gpu restricts it to what a CPU box can do: load, headroom, and reference-vs-oracle ids (a self-check of this tool's plumbing).
Checker tooling: imports oracle/ and, optionally, the reference; never part of the product path."""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CODE2LANG = {0: "zh", 1: "ja", 2: "en"}          # macros.py:15-19


def count_layers(sd) -> int:
    return 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("ar_decoder.layers."))


def load_presets(preset_dir, limit):
    out = []
    for p in sorted(glob.glob(os.path.join(preset_dir, "*.npz")))[:limit]:
        d = np.load(p)
        out.append(dict(name=os.path.splitext(os.path.basename(p))[0], audio=np.asarray(d["audio_tokens"]).astype(np.int64),
                        text=np.asarray(d["text_tokens"]).astype(np.int64), lang=CODE2LANG[int(d["lang_code"])]))
    return out


def job_inputs(preset, n_text, seed):
    from oracle import synth
    txt = synth.synth_text(n_text, seed)[None]
    text = np.concatenate([preset["text"], txt], -1)
    return preset["audio"], preset["text"], text, preset["lang"], preset["lang"]


def run_reference(ref_root, sd, num_layers, inputs, top_k, frames, useed):
    """the reference's own VALLE.inference on the CPU (oracle/make_golden.run_reference with explicit weights)"""
    from oracle import make_golden as MG
    MG.REF = ref_root
    c = dict(num_layers=num_layers, top_k=top_k, force_eos_at=frames, useed=useed, seed=0, eos_gain=1.0)
    return MG.run_reference(c, inputs=inputs, sd=sd)["codes"]


def first_diff(a, b):
    if a.shape != b.shape:
        return dict(equal=False, shape=[list(a.shape), list(b.shape)])
    d = np.argwhere(a != b)
    return dict(equal=not len(d), differing_ids=int(len(d)), first=[int(v) for v in d[0]] if len(d) else None)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--vocos")
    ap.add_argument("--vocos-config", help="config.yaml of charactr/vocos-encodec-24khz (for the pip `vocos` comparison)")
    ap.add_argument("--reference", help="checkout of Plachtaa/VALL-E-X to import and run on the CPU")
    ap.add_argument("--presets", help="directory of preset .npz prompts (default: <reference>/presets, else tests/golden/presets)")
    ap.add_argument("--max-presets", type=int, default=6)
    ap.add_argument("--n-text", type=int, default=60)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--ariths", default="f16x2,f32")
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--allow-unsafe-pickle", action="store_true",
                    help="load files the weights-only unpickler refuses with weights_only=False (executes code from the file: trusted origin only)")
    ap.add_argument("--out")
    args = ap.parse_args()

    import torch
    import vallex_amd  # noqa: F401
    from oracle import synth
    from oracle.vallex_oracle import VallexOracle, VocosOracle
    from vallex_amd.models.vallex import VALLE, expected_keys
    from vallex_amd.utils import generation as G

    rep = dict(ckpt=os.path.abspath(args.ckpt), torch=torch.__version__)
    ok = True
    # ---- load ----------------------------------------------------------------------------------------------------------
    t0 = time.time()
    ck = G._torch_load(args.ckpt, args.allow_unsafe_pickle or None)
    sd_t = ck["model"]
    nl = count_layers(sd_t)
    m = VALLE(1024, 16, nl, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
              prepend_bos=True, num_quantizers=8, engine_max_batch=2, engine_max_text=512, engine_max_prompt=1200,
              engine_max_new=max(args.frames, 64) + 8)
    m.load_state_dict(sd_t, strict=True)                        # raises like the reference on a missing / unexpected key
    sd = m._sd
    rep["load"] = dict(strict=True, keys=len(sd), expected_keys=len(expected_keys(nl)), num_layers=nl,
                       checkpoint_top_level=sorted(str(k) for k in ck), dtypes=sorted({str(v.dtype) for v in sd_t.values()}),
                       seconds=round(time.time() - t0, 1))
    # ---- presets -------------------------------------------------------------------------------------------------------
    pdir = args.presets or (os.path.join(args.reference, "presets") if args.reference else os.path.join(ROOT, "tests", "golden", "presets"))
    presets = load_presets(pdir, args.max_presets)
    if not presets:
        sys.exit(f"no .npz presets under {pdir}")
    rep["presets"] = dict(dir=os.path.abspath(pdir), used=[p["name"] for p in presets])
    jobs = []
    for i, p in enumerate(presets):
        inp = job_inputs(p, args.n_text, 100 + i)
        jobs.append(dict(name=p["name"] + "/greedy", inputs=inp, top_k=1, useed=None))
        jobs.append(dict(name=p["name"] + "/topk10", inputs=inp, top_k=10, useed=1234 + i))
    # ---- headroom (CPU oracle) -----------------------------------------------------------------------------------------
    orc = VallexOracle(sd, nl)
    orc.stats = {}
    a, t, text, pl, langs = jobs[0]["inputs"]
    with torch.no_grad():
        orc.ar_prefill(torch.from_numpy(text[0]), torch.from_numpy(a[0, :, 0]), t.shape[-1], pl, langs)
        orc._nar_stack(torch.randn(text.shape[-1] + a.shape[1] + args.frames, 1024) * 2.0, orc.w["nar_stage_embeddings.0.word_embeddings.weight"])
    mx = max(orc.stats.values())
    rep["headroom"] = dict(limit=2047.0, max_abs_operand={k: round(v, 2) for k, v in orc.stats.items()}, worst=round(mx, 2),
                           factor_below_limit=round(2047.0 / mx, 2) if mx > 0 else None,
                           expect_fp32_fallbacks=bool(mx >= 2047.0))
    orc.stats = None
    # ---- reference ids -------------------------------------------------------------------------------------------------
    ref_ids = {}
    if args.reference:
        for j in jobs:
            t0 = time.time()
            ref_ids[j["name"]] = run_reference(args.reference, sd, nl, j["inputs"], j["top_k"], args.frames, j["useed"])
            print(f"[reference] {j['name']}: {ref_ids[j['name']].shape} in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    # ---- engine --------------------------------------------------------------------------------------------------------
    rep["parity"] = {}
    eng_ids = {}
    if args.no_gpu:
        if args.reference:          # plumbing self-check: the CPU oracle against the reference
            res = {}
            for j in jobs:
                a, t, text, pl, langs = j["inputs"]
                us = None if j["useed"] is None else synth.uniforms(4096, 1, j["useed"])[:, 0]
                out = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=j["top_k"], prompt_language=pl,
                                    text_language=langs, uniforms=us, force_eos_at=args.frames)
                res[j["name"]] = first_diff(out, ref_ids[j["name"]])
                ok &= res[j["name"]]["equal"]
            rep["parity"]["oracle_vs_reference (no GPU)"] = res
    else:
        vsd = G._torch_load(args.vocos, args.allow_unsafe_pickle or None) if args.vocos else None
        for arith in args.ariths.split(","):
            m.engine_opts["arith"] = arith
            m._engine = None
            if vsd is not None:
                m.load_vocos_state_dict(vsd)
            m.to("cuda:0")
            res = {}
            for j in jobs:
                a, t, text, pl, langs = j["inputs"]
                us = None if j["useed"] is None else synth.uniforms(4096, 1, j["useed"])[:, 0]
                out = m.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=j["top_k"], prompt_language=pl,
                                  text_language=langs, uniforms=us, force_eos_at=args.frames)
                out = np.asarray(out)
                eng_ids[(arith, j["name"])] = out
                r = dict(frames=int(out.shape[1]), fallbacks=m.engine.last_fallbacks())
                if args.reference:
                    r["vs_reference"] = first_diff(out, ref_ids[j["name"]])
                    ok &= r["vs_reference"]["equal"]
                res[j["name"]] = r
            rep["parity"][arith] = res
        ar = args.ariths.split(",")
        if not args.reference and len(ar) > 1:
            rep["parity"]["cross_arith"] = {j["name"]: first_diff(eng_ids[(ar[0], j["name"])], eng_ids[(ar[1], j["name"])]) for j in jobs}
            ok &= all(v["equal"] for v in rep["parity"]["cross_arith"].values())
        # ---- Vocos ------------------------------------------------------------------------------------------------------
        if vsd is not None:
            codes = eng_ids[(ar[-1], jobs[0]["name"])]
            wav = m.engine.vocos_decode([codes[0]], 2)[0]
            v = dict(samples=int(wav.shape[0]))
            wref = VocosOracle(m._vocos_sd).decode_codes(codes, 2)[0]
            v["rms_vs_cpu_restatement"] = float(np.sqrt(np.mean((wav - wref) ** 2)))
            ok &= v["rms_vs_cpu_restatement"] <= 1e-4
            try:
                from vocos import Vocos                                   # the pip package (utils/generation.py:4,89)
                if not args.vocos_config:
                    raise RuntimeError("pass --vocos-config config.yaml to compare against the pip package")
                pk = Vocos.from_hparams(args.vocos_config)
                pk.load_state_dict(vsd, strict=False)
                pk.eval()
                with torch.no_grad():
                    fr = torch.from_numpy(codes).permute(2, 0, 1)        # (8, 1, T)  utils/generation.py:148
                    wp = pk.decode(pk.codes_to_features(fr), bandwidth_id=torch.tensor([2])).squeeze().numpy()
                v["rms_vs_pip_vocos"] = float(np.sqrt(np.mean((wav - wp) ** 2)))
                ok &= v["rms_vs_pip_vocos"] <= 1e-4
            except Exception as e:
                v["pip_vocos"] = f"not compared: {type(e).__name__}: {e}"
            rep["vocos"] = v
    rep["ok"] = bool(ok)
    txt = json.dumps(rep, indent=1)
    if args.out:
        open(args.out, "w").write(txt)
    print(txt)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
