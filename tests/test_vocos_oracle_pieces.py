"""CPU: independent checks of the pieces of the Vocos restatement (oracle/vallex_oracle.py VocosOracle).

The pip `vocos` package and its weights are absent offline, so the restatement as a whole stays "parity unpinned" (DESIGN.md
section 2).  What CAN be pinned here is that every piece computes what the torch building blocks the package is made of compute:
  * backbone: the same synthetic weights loaded into torch.nn modules (Conv1d, LayerNorm, Embedding, Linear, GELU) wired like
    vocos.models.VocosBackbone / vocos.modules.ConvNeXtBlock / AdaLayerNorm (recalled structure);
  * codes_to_features: nn.Embedding + offsets, vocos.pretrained.Vocos.codes_to_features;
  * ISTFT head ("same" padding): torch.istft (center=True) on the interior samples, where the two differ only by the 160-sample
    shift between trimming (n_fft - hop)/2 = 480 and n_fft/2 = 640;
  * ISTFT head, EVERY sample incl. the edges, against a third-party port of the package's own head code: the installed
    `transformers` ships `Xcodec2ISTFTHead` (models/xcodec2/modeling_xcodec2.py), which its docstring declares to be the
    "same"-padding ISTFT of vocos/spectral_ops.py (gemelo-ai/vocos@c859e3b) behind the Linear -> exp/clip(1e2) -> polar head;
    built with Vocos-EnCodec's geometry (dim 384, n_fft 1280, hop 320) and our head weights it IS vocos.heads.ISTFTHead.
"""
import numpy as np
import torch
import torch.nn as nn

from oracle import synth
from oracle.vallex_oracle import VocosOracle

C, H, NB = synth.VOCOS_DIM, synth.VOCOS_IDIM, synth.VOCOS_NFFT + 2


class AdaLN(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale = nn.Embedding(4, C)
        self.shift = nn.Embedding(4, C)
        self.ln = nn.LayerNorm(C, eps=1e-6, elementwise_affine=False)

    def forward(self, x, bid):
        return self.ln(x) * self.scale(bid) + self.shift(bid)


class Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.dwconv = nn.Conv1d(C, C, 7, padding=3, groups=C)
        self.norm = AdaLN()
        self.pwconv1 = nn.Linear(C, H)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(H, C)
        self.gamma = nn.Parameter(torch.zeros(C))

    def forward(self, x, bid):
        r = x
        x = self.dwconv(x).transpose(1, 2)
        x = self.pwconv2(self.act(self.pwconv1(self.norm(x, bid))))
        return r + (self.gamma * x).transpose(1, 2)


class Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Conv1d(synth.VOCOS_INCH, C, 7, padding=3)
        self.norm = AdaLN()
        self.convnext = nn.ModuleList([Block() for _ in range(synth.VOCOS_LAYERS)])
        self.final_layer_norm = nn.LayerNorm(C, eps=1e-6)

    def forward(self, feat, bid):
        x = self.embed(feat)
        x = self.norm(x.transpose(1, 2), bid).transpose(1, 2)
        for blk in self.convnext:
            x = blk(x, bid)
        return self.final_layer_norm(x.transpose(1, 2))


def _load(mod, sd):
    own = mod.state_dict()
    ren = {}
    for k, v in sd.items():
        if not k.startswith("backbone."):
            continue
        kk = k[len("backbone."):].replace("norm.scale.weight", "norm.scale.weight").replace("norm.shift.weight", "norm.shift.weight")
        ren[kk] = torch.from_numpy(v)
    missing = [k for k in own if k not in ren]
    assert not missing, missing
    mod.load_state_dict({k: ren[k] for k in own}, strict=True)


def test_backbone_equals_torch_modules():
    sd = synth.vocos_state_dict(2)
    orc = VocosOracle(sd)
    bb = Backbone().eval()
    _load(bb, sd)
    feat = torch.randn(2, synth.VOCOS_INCH, 53, generator=torch.Generator().manual_seed(1))
    for bid in (0, 2):
        with torch.no_grad():
            ref = bb(feat, torch.tensor([bid]))
            got = orc.backbone(feat, bid)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


def test_codes_to_features_equals_embedding_sum():
    sd = synth.vocos_state_dict(2)
    orc = VocosOracle(sd)
    emb = nn.Embedding.from_pretrained(torch.from_numpy(sd["feature_extractor.codebook_weights"]))
    codes = torch.randint(0, 1024, (8, 2, 31), generator=torch.Generator().manual_seed(3))
    ref = sum(emb(codes[q] + 1024 * q) for q in range(8)).transpose(1, 2)
    np.testing.assert_allclose(orc.codes_to_features(codes).numpy(), ref.numpy(), atol=1e-6, rtol=0)


def test_istft_head_equals_torch_istft_on_interior_samples():
    sd = synth.vocos_state_dict(2)
    orc = VocosOracle(sd)
    n_fft, hop = synth.VOCOS_NFFT, synth.VOCOS_HOP
    T = 40
    x = torch.randn(1, T, C, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        got = orc.head(x)[0]                                            # (320 T,)
        o = torch.nn.functional.linear(x, orc.w["head.out.weight"], orc.w["head.out.bias"]).transpose(1, 2)
        mag, p = o.chunk(2, dim=1)
        S = torch.clip(torch.exp(mag), max=1e2) * (torch.cos(p) + 1j * torch.sin(p))
        ref = torch.istft(S, n_fft, hop, n_fft, torch.hann_window(n_fft), center=True)[0]   # (320 (T-1),), trims 640 per side
    assert got.shape[0] == hop * T and ref.shape[0] == hop * (T - 1)
    shift = n_fft // 2 - (n_fft - hop) // 2                             # 160
    lo, hi = n_fft, hop * (T - 1) - n_fft                               # fully overlapped region of both
    np.testing.assert_allclose(got[shift + lo: shift + hi].numpy(), ref[lo:hi].numpy(), atol=2e-5, rtol=1e-5)


def test_istft_head_equals_transformers_port_of_the_vocos_head():
    """the head of the restatement == transformers' port of vocos.heads.ISTFTHead + vocos.spectral_ops.ISTFT(padding="same"),
    every output sample (edges included), also with magnitudes that hit the exp clip at 1e2"""
    import types
    from transformers.models.xcodec2.modeling_xcodec2 import Xcodec2ISTFTHead
    sd = synth.vocos_state_dict(2)
    orc = VocosOracle(sd)
    port = Xcodec2ISTFTHead(types.SimpleNamespace(hidden_size=C, n_fft=synth.VOCOS_NFFT, hop_length=synth.VOCOS_HOP)).eval()
    with torch.no_grad():
        port.linear.weight.copy_(orc.w["head.out.weight"])
        port.linear.bias.copy_(orc.w["head.out.bias"])
    for T, gain in ((1, 1.0), (2, 1.0), (3, 1.0), (40, 1.0), (600, 1.0), (40, 400.0)):
        x = gain * torch.randn(2, T, C, generator=torch.Generator().manual_seed(10 + T))
        with torch.no_grad():
            ref = port(x)[:, 0]                                          # (B, 320 T)
            got = orc.head(x)
        assert got.shape == ref.shape == (2, synth.VOCOS_HOP * T)
        scale = float(ref.abs().max())
        if gain > 1:                                                     # the clip is really exercised
            o = torch.nn.functional.linear(x, orc.w["head.out.weight"], orc.w["head.out.bias"])
            assert float(o[..., : NB // 2].max()) > np.log(1e2)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-6 * max(scale, 1.0), rtol=1e-5)


def test_convnext_branch_matches_a_third_party_convnext_1d_block():
    """The ConvNeXt block of the backbone against an implementation that is not ours and not recalled: the installed `transformers`
    ships `Qwen3OmniMoeConvNeXtBlock` (models/qwen3_omni_moe, the code2wav vocoder), the 1-D ConvNeXt block of the same lineage --
    depthwise Conv1d k = 7 -> LayerNorm(eps 1e-6) over channels -> Linear(C, 4C) -> GELU (erf) -> Linear(4C, C) -> gamma -> + input.
    Differences that the comparison removes: its depthwise conv is CAUSAL (6 frames of left padding) where Vocos centres it
    (padding 3), so its branch output at frame t + 3 is the centred one at t (interior frames); its LayerNorm is a plain affine
    one, which equals AdaLayerNorm with weight = scale[id], bias = shift[id]; its hidden width is 4C (the oracle is shape-agnostic).
    Pins the operation order, the LayerNorm epsilon, the exact GELU and the placement of gamma of `VocosOracle.convnext_branch`."""
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeConvNeXtBlock
    torch.manual_seed(5)
    Cq, T, bid = synth.VOCOS_DIM, 40, 2
    blk = Qwen3OmniMoeConvNeXtBlock(Cq).eval()
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(torch.randn_like(prm) * 0.2)
        blk.norm.weight.copy_(1.0 + 0.1 * torch.randn(Cq))
        blk.gamma.copy_(0.3 * torch.randn(Cq))
    scale = torch.zeros(4, Cq)
    shift = torch.zeros(4, Cq)
    scale[bid], shift[bid] = blk.norm.weight.detach(), blk.norm.bias.detach()
    sd = {"backbone.convnext.0.dwconv.weight": blk.dwconv.conv.weight.detach().numpy(),
          "backbone.convnext.0.dwconv.bias": blk.dwconv.conv.bias.detach().numpy(),
          "backbone.convnext.0.norm.scale.weight": scale.numpy(), "backbone.convnext.0.norm.shift.weight": shift.numpy(),
          "backbone.convnext.0.pwconv1.weight": blk.pwconv1.weight.detach().numpy(),
          "backbone.convnext.0.pwconv1.bias": blk.pwconv1.bias.detach().numpy(),
          "backbone.convnext.0.pwconv2.weight": blk.pwconv2.weight.detach().numpy(),
          "backbone.convnext.0.pwconv2.bias": blk.pwconv2.bias.detach().numpy(),
          "backbone.convnext.0.gamma": blk.gamma.detach().numpy()}
    orc = VocosOracle(sd)
    x = torch.randn(2, Cq, T)
    with torch.no_grad():
        theirs = blk(x) - x                                   # residual branch, causal: frame t + 3 <-> centred frame t
        ours = orc.convnext_branch(x, 0, bid)
    np.testing.assert_allclose(ours[:, :, 3: T - 3].numpy(), theirs[:, :, 6:].numpy(), atol=2e-5, rtol=0)
    assert float(ours.abs().max()) > 0.1                      # a non-trivial branch
