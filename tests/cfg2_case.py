"""BASELINE config 2 at its own shape, as a pure function of seeds: ConformerLangModel 12-layer d256, 14 languages, batch 64 of
3 s utterances (F = 301 frames with pad 16, T = 151 after subsampling).  Used by ``oracle/gen_golden_r2.py`` (which feeds the
REFERENCE with it in the build container and stores the outputs in tests/golden/cfg2_step.npz) and by the GPU test that runs
the same step on the HIP engine.  The 46 M weights are never stored: both sides rebuild them from the seed."""
import torch

LANGS = [f"l{k:02d}" for k in range(14)]
L2V = {l: 40 for l in LANGS}
L2I = {l: k for k, l in enumerate(LANGS)}
DIMS = dict(n_blocks=12, encoder_dim=256, dim_head=64, heads=4, last_dim_head=32)
LANG = "l03"
B, SECONDS, TEXT_LEN = 64, 3.0, 20
SAMPLE = 2048                     # gradient elements kept per tensor in the fixture


def product_cfg(**kw):
    from lidk.layout import ConformerCfg
    return ConformerCfg(lang2vocab=dict(L2V), lang2index=dict(L2I), hidden_dim=32, dropout=0.0, pos_dropout=0.0, **DIMS, **kw)


def weights():
    """name -> CPU f32 tensor (reference state_dict keys): torch.nn's own initialisers in the reference's construction order
    under manual_seed(0) (lidk.layout.init_values), then LayerNorm / BatchNorm affine parameters moved off 1 / 0 and the
    BatchNorm running statistics off 0 / 1 so that they matter."""
    from lidk.layout import init_values, model_specs
    torch.manual_seed(0)
    cfg = product_cfg()
    sd = init_values(cfg)
    g = torch.Generator().manual_seed(7)
    for name in sorted(sd):
        v = sd[name]
        if v.dim() == 1 and (".norm." in name or "post_norm." in name or ".conv.net.0." in name or ".conv.net.5." in name):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
    _, buffers, _, _ = model_specs(cfg)
    for name, shape, dt in buffers:
        if name.endswith("running_mean"):
            sd[name] = 0.2 * torch.randn(shape, generator=g)
        elif name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=g)
        else:
            sd[name] = torch.zeros(shape, dtype=dt)
    return sd


def batch():
    """-> (mel (64, 301, 80) f32 dB, texts (64, 20) int64): 64 utterances of language l03 from the learnable synthetic corpus
    (tone-pair transcripts), features by the CPU oracle (normalize -> log-mel pad 16, no augmentation)."""
    from lid.raw_datasets import SyntheticMergedDataset
    from oracle import features as of
    ds = SyntheticMergedDataset(False, L2I, L2V, items_per_lang=B, seconds=SECONDS, text_len=TEXT_LEN, seed=4242,
                                transcript="tones", type="mel", pad=16)
    base = L2I[LANG] * B
    wav = torch.stack([ds.waveform(base + i) for i in range(B)])
    texts = torch.stack([ds.text(base + i) for i in range(B)])
    mel = of.wav2mel(of.normalize_wav(wav), pad=16).transpose(1, 2).contiguous()
    return mel, texts


def sample_index(name: str, numel: int) -> torch.Tensor:
    """Seeded element subset of a gradient tensor (all of it when it has at most SAMPLE elements)."""
    if numel <= SAMPLE:
        return torch.arange(numel)
    seed = sum((i + 1) * ord(c) for i, c in enumerate(name)) % (2 ** 31)
    return torch.randperm(numel, generator=torch.Generator().manual_seed(seed))[:SAMPLE].sort().values
