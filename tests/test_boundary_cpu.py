"""Drop-in boundary: the import block of the reference launcher (lid/main.py:8-21) resolves against this package,
and a YAML in the reference's conf schema builds module + datasets + samplers + CometLogger(**cfg.logger.comet) +
Trainer(**cfg.trainer) the way the reference's ``main`` does (lid/main.py:29-135).  No GPU work: nothing is trained here
(tests/test_gpu_train.py::test_reference_schema_yaml_trains_from_disk runs the same YAML through ``fit`` on the GPU)."""
import os

import pytest
import torch

from conftest import PKG
import ref_schema


def test_reference_launcher_import_block_resolves():
    from ccml.callbacks.ckpt_callback import CkptCallback  # noqa: F401
    from ccml.callbacks.lr_callback import LrCallback  # noqa: F401
    from ccml.callbacks.profile_callback import ProfileCallback  # noqa: F401
    from ccml import seed_everything  # noqa: F401
    from ccml.trainer import Trainer  # noqa: F401
    from lid.LidModule_ASR import LidModule  # noqa: F401
    from lid.LidModule_ASR_Supervised import LidSuperviseModule  # noqa: F401
    from lid.raw_datasets import MergedDataset, MutiBatchSampler  # noqa: F401
    from lid.tokenizer import CTCTokenizer  # noqa: F401
    from ccml.loggers.wandb_logger import WandbLogger  # noqa: F401
    from ccml.loggers.comet_logger import CometLogger  # noqa: F401


def test_reference_schema_yaml_builds_module_loggers_and_trainer(tmp_path, monkeypatch):
    from collections import defaultdict
    from ccml.callbacks.ckpt_callback import CkptCallback
    from ccml.callbacks.lr_callback import LrCallback
    from ccml.callbacks.profile_callback import ProfileCallback
    from ccml.loggers.comet_logger import CometLogger
    from ccml.trainer import Trainer
    from lid import hydra_lite
    from lid.LidModule_ASR_Supervised import LidSuperviseModule
    from lid.raw_datasets import MergedDataset, MutiBatchSampler
    from lid.tokenizer import CTCTokenizer

    monkeypatch.chdir(tmp_path)
    corpus = ref_schema.make_corpus(str(tmp_path), n_train=6, n_val=2)
    ref_schema.write_yaml(str(tmp_path / "conf"), corpus)
    cfg = hydra_lite.load_config(str(tmp_path / "conf"), "xf_like", ["trainer.total_epoch=3"])

    # interpolation + YAML anchor/alias: logger names follow model.model_name, which itself interpolates three other keys
    assert cfg["model"]["model_name"] == "lr_0.01_dr_0.1_bs_4_conform_True"
    assert cfg["logger"]["comet"]["name"] == cfg["model"]["model_name"] == cfg["logger"]["wandb"]["name"]
    assert cfg["supervised"] is True and cfg["trainer"]["backend"] == "nccl" and cfg["trainer"]["total_epoch"] == 3

    # ---- what the reference's main() does with the config, in its order
    train_conf, model_conf, module_conf, data_conf = cfg["trainer"], cfg["model"], cfg["module"], cfg["data"]
    comet_conf = cfg["logger"]["comet"]
    tokenizers, lang2index, lang2vocab = defaultdict(dict), defaultdict(int), defaultdict(int)
    manifests = {"train": [], "val": [], "test": []}
    for item in data_conf["langs"]:
        tokenizers[item["lang"]] = CTCTokenizer(item["vocab"])               # vocab is a FILE PATH in the reference's confs
        for split in manifests:
            manifests[split].append(item[f"{split}_manifest"])
        lang2index[item["lang"]] = item["id"]
        lang2vocab[item["lang"]] = len(tokenizers[item["lang"]].export_vocab())
    assert dict(lang2vocab) == {lang: 12 for lang in ref_schema.LANGS}
    module = LidSuperviseModule(**module_conf, **model_conf, lang2vocab=lang2vocab, lang2index_dict=lang2index,
                                tokenizer_dict=tokenizers)               # extra keys (model_name, freeze_*) are swallowed
    assert module.hyper_parameters["encoder_dim"] == 64 and module.model.cfg.n_blocks == 2
    sets = {}
    for split in manifests:
        sets[split] = MergedDataset(train=split == "train", manifest_files=manifests[split], lang2index_dict=lang2index,
                                    lang2tokenizer=tokenizers, max_duration=data_conf["max_duration"] if split == "train" else 16.7,
                                    source=data_conf["source"], **data_conf["feature"])
    assert len(sets["train"]) == 18 and len(sets["val"]) == 6
    wav, text, path, lang = sets["train"][0]
    assert wav.dim() == 2 and wav.shape[0] == 1 and text.dtype == torch.int64 and lang == "Persian" and os.path.exists(path)
    params = dict(data_conf["dataloader_params"])
    sc = data_conf["sampler_common"]
    params["train_batch_sampler"] = MutiBatchSampler(sets["train"].samplers, batch_size=sc["train_batch_size"], drop_last=True)
    params["val_batch_sampler"] = MutiBatchSampler(sets["val"].samplers, batch_size=sc["val_batch_size"], drop_last=False)
    params["test_batch_sampler"] = MutiBatchSampler(sets["test"].samplers, batch_size=sc["test_batch_size"], drop_last=False)
    batch = sets["train"].collate_fn([sets["train"][i] for i in next(iter(params["train_batch_sampler"]))])
    assert len(batch) == 6 and batch[1].shape[0] == 4 and len(set(batch[5].tolist())) == 1        # single-language batch

    comet = CometLogger(**comet_conf)                                    # api_key / project / entity / name keywords
    trainer = Trainer(callbacks=[CkptCallback(file_name_metric=["epoch", "val_loss"], save_topk=2), LrCallback(),
                                 ProfileCallback()], loggers=[comet], **train_conf)
    assert trainer.device.type == "cpu" and trainer.accumulate_grad == 2 and trainer.total_epoch == 3 and not trainer.ddp
    comet.log({"loss": 1.0})
    assert os.path.exists(comet.path)


def test_unsupervised_flag_selects_the_backbone_module():
    """`supervised: false` routes to lid.LidModule_ASR.LidModule exactly as the reference launcher does."""
    import lid.main as launcher
    from lid.LidModule_ASR import LidModule
    assert launcher.LidModule is LidModule


def test_own_launcher_builds_from_reference_schema_yaml(tmp_path, monkeypatch):
    """This repository's lid/main.py::build consumes the same YAML (file-path vocab, xf manifests) without overrides."""
    import lid.main as launcher
    from lid import hydra_lite
    monkeypatch.chdir(tmp_path)
    corpus = ref_schema.make_corpus(str(tmp_path), n_train=5, n_val=2)
    ref_schema.write_yaml(str(tmp_path / "conf"), corpus)
    cfg = hydra_lite.load_config(str(tmp_path / "conf"), "xf_like")
    module, sets, params = launcher.build(cfg)
    assert type(module).__name__ == "LidSuperviseModule" and len(sets["train"]) == 15
    assert len(params["train_batch_sampler"]) == 3                       # 5 // 4 per language, drop_last
    assert os.path.isdir(PKG)
