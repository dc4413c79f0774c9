"""Host-side parts of the WavLM path (no GPU): the span-mask draw against the reference's compute_mask_indices (KATs captured under
fixed numpy seeds in tests/golden/wavlm_model.npz), state-dict naming, LidModule construction from the reference's YAML keys."""
import numpy as np
import pytest
import torch

from conftest import load_npz
import wavlm_case as wc


def test_span_mask_matches_reference_compute_mask_indices():
    from lidk.wavlm import span_mask
    g = load_npz("wavlm_model.npz")
    for case in range(3):
        rows, size, prob, length, mm, seed = g[f"mask{case}_args"]
        np.random.seed(int(seed))
        got = span_mask((int(rows), int(size)), None, float(prob), int(length), min_masks=int(mm))
        assert np.array_equal(got, g[f"mask{case}"]), case
    pm = torch.zeros(3, 49, dtype=torch.bool)
    pm[1, 40:] = True
    np.random.seed(9)
    got = span_mask((3, 49), pm, 0.3, 5, min_masks=2)
    assert np.array_equal(got, g["mask_pad"]) and not got[1, 40:].any()


def test_relative_bucket_table_matches_reference_bias():
    from lidk.wavlm import relative_buckets
    g = load_npz("wavlm_fwd.npz")
    emb = wc.backbone_weights()["encoder.layers.0.self_attn.relative_attention_bias.weight"]
    T = g["pos_bias0"].shape[-1]
    i, j = torch.arange(T)[:, None], torch.arange(T)[None, :]
    table = emb[relative_buckets(j - i, 320, 800)].permute(2, 0, 1)
    assert float((table - torch.from_numpy(g["pos_bias0"])).abs().max()) == 0.0


def test_wavlm_model_state_dict_names_and_module_surface():
    from lid.LidModule_ASR import LidModule
    from lid.tokenizer import CTCTokenizer
    from lidk.wavlm import WavLMBackbone
    toks = {k: CTCTokenizer([chr(0x4E00 + i) for i in range(v)]) for k, v in wc.L2V.items()}
    mod = LidModule(optimizer_name="adam", optimizer_param={"lr": 1e-4, "weight_decay": 1e-6}, scheduler="tristage", interval=50,
                    freeze_tranformer_epoch=1, freeze_encoder_epoch=100, froze_wav2vec_model_epoch=-1, use_wav2vec=False,
                    conformer_linear=True, extrme_mode=True, sr=16000, dropout=0.1, linear_dim=768, mask=True, num_layers=1,
                    hidden_dim=32, model_name="x", feature_selection="last_hidden_state", use_pre_train=True, mask_channel_prob=0.15,
                    double_swish=False, mask_prob=0.15, use_mask=False, dim_head=32, num_head=8, lang2vocab=wc.L2V,
                    lang2index_dict=wc.L2I, tokenizer_dict=toks, wavlm_cfg=wc.CFG)
    sd = mod.model.state_dict()
    want = {"model.featurizer.model." + k for k in WavLMBackbone.param_shapes(wc.CFG)} | set(wc.head_weights())
    assert set(sd) == want
    assert mod.model.backbone.cfg["mask_prob"] == 0.15 and mod.model.backbone.cfg["mask_channel_prob"] == 0.15
    assert set(WavLMBackbone.param_shapes(wc.CFG)) == set(wc.backbone_shapes(2))
    trainable = [n for n, p in mod.model.named_parameters() if p.requires_grad]
    # "frozen" backbone = what the reference's freeze_* helpers leave trainable: WavLM's layer_norm + mask_emb (never frozen)
    assert {n for n in trainable if n.startswith("model.featurizer.")} == {
        "model.featurizer.model." + k for k in ("layer_norm.weight", "layer_norm.bias", "mask_emb")}
    mod.model.unfreeze_tranformer_encoder()                      # lid/WavLMMutiLangModel.py:106-112: exactly the encoder.* parameters
    now = {n for n, p in mod.model.named_parameters() if p.requires_grad} - set(trainable)
    assert now == {"model.featurizer.model." + k for k in WavLMBackbone.param_shapes(wc.CFG) if k.startswith("encoder.")}
    mod.model.freeze_tranformer_encoder()
    assert [n for n, p in mod.model.named_parameters() if p.requires_grad] == trainable
    mod.model.unfreeze_feature_extractor()                          # lid/WavLMMutiLangModel.py:86-94: extractor + post_extract_proj
    now = {n for n, p in mod.model.named_parameters() if p.requires_grad} - set(trainable)
    assert now and all(n.startswith(("model.featurizer.model.feature_extractor.", "model.featurizer.model.post_extract_proj."))
                       for n in now)
    mod.model.freeze_feature_extractor()
    assert [n for n, p in mod.model.named_parameters() if p.requires_grad] == trainable
    with pytest.raises(ValueError):                                 # a wav2vec2 model needs a checkpoint or a config
        LidModule(lang2vocab=wc.L2V, lang2index_dict=wc.L2I, tokenizer_dict=toks, use_wav2vec=True, conformer_linear=True)


def test_wav2vec2_model_state_dict_names_and_module_surface():
    """LidModule(use_wav2vec=True) builds the wav2vec2-backbone model (lid/Wav2vecMutiLangModel.py): fairseq parameter names under
    model.featurizer.upstream.model.*, the s3prl Featurizer's mixing weights, no relative-position parameters; the freeze_*
    helpers leave layer_norm + mask_emb (+ the mixing weights and heads) trainable, like the reference's."""
    from lid.LidModule_ASR import LidModule
    from lid.tokenizer import CTCTokenizer
    from lidk.wavlm import WavLMBackbone
    toks = {k: CTCTokenizer([chr(0x4E00 + i) for i in range(v)]) for k, v in wc.L2V.items()}
    mod = LidModule(optimizer_name="adam", optimizer_param={"lr": 1e-4}, scheduler="none", use_wav2vec=True, conformer_linear=True,
                    sr=16000, dropout=0.0, linear_dim=768, mask=True, hidden_dim=32, feature_selection="hidden_states", dim_head=32,
                    num_head=8, lang2vocab=wc.L2V, lang2index_dict=wc.L2I, tokenizer_dict=toks, wav2vec_cfg=wc.W2V_CFG)
    m = mod.model
    pre = "model.featurizer.upstream.model."
    shapes = WavLMBackbone.param_shapes(wc.W2V_CFG)
    assert not any("grep" in k or "relative_attention_bias" in k for k in shapes)
    want = {pre + k for k in shapes} | set(wc.head_weights()) | {"model.featurizer.weights"}
    assert set(m.state_dict()) == want
    assert m.state_dict()["model.featurizer.weights"].shape == (wc.W2V_CFG["encoder_layers"] + 1,)
    # load_wav2vec2_for_finetune's fixed masking (wav2vec2_expert.py:210-212) and LayerDrop off for the hidden-state mix (:205-207)
    bcfg = m.backbone.cfg
    assert m.backbone.pad_mask and not m.backbone.rel_pos and (bcfg["mask_prob"], bcfg["mask_channel_prob"]) == (0.2, 0.2)
    assert bcfg["mask_channel_length"] == 64 and bcfg["encoder_layerdrop"] == -1.0
    trainable = {n for n, p in m.named_parameters() if p.requires_grad and n.startswith("model.featurizer.")}
    assert trainable == {pre + "layer_norm.weight", pre + "layer_norm.bias", pre + "mask_emb", "model.featurizer.weights"}
    m.unfreeze_tranformer_encoder()
    now = {n for n, p in m.named_parameters() if p.requires_grad and n.startswith(pre)}
    assert {pre + k for k in shapes if k.startswith("encoder.")} <= now
    m.froze_wav2vec_model()
    assert not any(p.requires_grad for p in m.model.parameters()) and all(p.requires_grad for p in m.lang_discriminator.parameters())
    m.unfroze_wav2vec_model()
    last = LidModule(use_wav2vec=True, conformer_linear=True, feature_selection="last_hidden_state", lang2vocab=wc.L2V,
                     lang2index_dict=wc.L2I, tokenizer_dict=toks, wav2vec_cfg=wc.W2V_CFG, linear_dim=768)
    assert "model.featurizer.weights" not in last.model.state_dict()
    assert last.model.backbone.cfg["encoder_layerdrop"] == 0.0                   # drop_layer: the checkpoint's own LayerDrop stays
    # the Large / XLS-R form and the nested fairseq cfg layout
    nested = {"model": {k: v for k, v in wc.XLSR_CFG.items() if k != "normalize"}, "task": {"normalize": True}}
    big = LidModule(use_wav2vec=True, conformer_linear=True, feature_selection="last_hidden_state", lang2vocab=wc.L2V,
                    lang2index_dict=wc.L2I, tokenizer_dict=toks, wav2vec_cfg=nested, linear_dim=1024, hidden_dim=64)
    bb = big.model.backbone
    assert bb.normalize and bb.pre_ln and bb.ln_extractor and bb.conv_bias and (bb.d, bb.H, bb.ffn) == (1024, 16, 4096)
    names = set(big.model.state_dict())
    for i in range(7):
        assert {pre + f"feature_extractor.conv_layers.{i}.0.bias", pre + f"feature_extractor.conv_layers.{i}.2.1.weight",
                pre + f"feature_extractor.conv_layers.{i}.2.1.bias"} <= names
    assert pre + "feature_extractor.conv_layers.0.2.weight" not in names


def test_wavlm_large_config_ignores_normalize_and_keep_last_lang_freezes_other_heads():
    """WavLM Large cfg (lid/conf/xf_asr_extra_finetune.yaml:12): layer_norm extractor without conv bias, pre-LN layers, the gate's
    parameters; ``normalize`` in the checkpoint cfg is NOT applied (lid/wavlm/example.py:43-45 never normalises).
    keep_last_lang_model_train (lid/WavLMMutiLangModel.py:114-123) switches requires_grad off for every other head."""
    from lid.WavLMMutiLangModel import WavLMMutiLangModel
    m = WavLMMutiLangModel(wavlm_cfg=dict(wc.WAVLM_LARGE_CFG), lang2vocab=wc.L2V, lang2index=wc.L2I, conformer_linear=True,
                           linear_dim=1024, hidden_dim=64)
    bb = m.backbone
    assert bb.pre_ln and bb.ln_extractor and not bb.conv_bias and bb.rel_pos and not bb.normalize
    assert "model.featurizer.model.feature_extractor.conv_layers.0.0.bias" not in m.state_dict()
    m.keep_last_lang_model_train("b")
    for n, p in m.named_parameters():
        if n.startswith("model.last_projects."):
            assert p.requires_grad == n.startswith("model.last_projects.b."), n


def test_padding_frames_closed_form_equals_forward_padding_mask():
    """WavLM.forward_padding_mask (lid/wavlm/WavLM.py:290-296) over the model's (B, L) sample mask against the closed form the
    backbone uses per step (frame t of an n-sample utterance is padding iff t * (L // T) >= n)."""
    rng = np.random.default_rng(0)
    for Lw, Tn in ((48000, 149), (16000, 49), (12345, 38)):
        n = np.concatenate([rng.integers(400, Lw + 1, size=9), [Lw]])
        per = Lw // Tn
        pm = torch.ones(len(n), Lw, dtype=torch.bool)
        for i, v in enumerate(n):
            pm[i, :v] = False
        brute = pm[:, :per * Tn].view(len(n), Tn, per).all(-1).numpy()
        closed = np.arange(Tn)[None, :] >= (-(-n.astype(np.int64) // per))[:, None]
        assert (brute == closed).all()
