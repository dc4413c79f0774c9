"""LidModule — the reference's CCMLModule for the pretrained-backbone LID models (lid/LidModule_ASR.py:17-409:
``WavLMMutiLangModel`` / ``Wav2vecMutiLangModel`` with per-language CTC heads).

The reference launcher imports this name unconditionally (lid/main.py:13) and builds it when the YAML says
``supervised: false``.  The backbones behind it are SURVEY 8f rows N1 (WavLM) / N2 (wav2vec2); until their HIP paths are
built this class refuses construction with a clear message instead of silently training something else — there is no
torch fallback for any model in this package.
"""
from ccml.ccml_module import CCMLModule


class LidModule(CCMLModule):
    def __init__(self, *args, use_wav2vec: bool = False, **kwargs):
        super().__init__(*args, use_wav2vec=use_wav2vec, **kwargs)
        backbone = "wav2vec2 (SURVEY 8f N2)" if use_wav2vec else "WavLM (SURVEY 8f N1)"
        raise NotImplementedError(
            f"LidModule: the {backbone} backbone has no HIP path in this build; the Conformer LID path is "
            "lid.LidModule_ASR_Supervised.LidSuperviseModule (YAML `supervised: true`)")
