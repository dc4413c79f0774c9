"""Training / test launcher with the reference's call shape (lid/main.py:23-147): build tokenizers, the CCMLModule, datasets,
single-language batch samplers, callbacks and the Trainer from the YAML, then ``fit`` or ``test``.

    python main.py --config-name synthetic_cfg1 trainer.total_epoch=2
    torchrun --nproc-per-node 8 main.py --config-name synthetic_cfg2        # data parallel: one process per GPU

``data.source: synthetic`` uses the built-in synthetic corpus; ``common_voice`` / ``xf_asr`` read manifests as the reference."""
import logging
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from ccml import seed_everything  # noqa: E402
from ccml.callbacks.ckpt_callback import CkptCallback  # noqa: E402
from ccml.callbacks.lr_callback import LrCallback  # noqa: E402
from ccml.callbacks.profile_callback import ProfileCallback  # noqa: E402
from ccml.loggers.comet_logger import CometLogger  # noqa: E402
from ccml.loggers.jsonl_logger import JsonlLogger  # noqa: E402
from ccml.loggers.wandb_logger import WandbLogger  # noqa: E402,F401
from ccml.trainer import Trainer  # noqa: E402
from lid.LidModule_ASR import LidModule  # noqa: E402
from lid.LidModule_ASR_Supervised import LidSuperviseModule  # noqa: E402
from lid.raw_datasets import MergedDataset, MutiBatchSampler, SyntheticMergedDataset  # noqa: E402
from lid.tokenizer import CTCTokenizer  # noqa: E402

try:
    import hydra
    hydra_main = hydra.main
except ImportError:                                  # Hydra is not in this image: same decorator shape, own loader
    from lid import hydra_lite
    hydra_main = hydra_lite.main


def build(cfg, rank=0, world=1):
    data = cfg["data"]
    tokenizers, lang2index, lang2vocab, manifests = {}, {}, {}, {"train": [], "val": [], "test": []}
    for item in data["langs"]:
        vocab = item["vocab"] if "vocab" in item else [chr(0x4E00 + i) for i in range(int(item["vocab_size"]))]
        tokenizers[item["lang"]] = CTCTokenizer(vocab)
        lang2index[item["lang"]] = item["id"]
        lang2vocab[item["lang"]] = len(tokenizers[item["lang"]].export_vocab())
        for split in manifests:
            manifests[split].append(item.get(f"{split}_manifest"))
    # `supervised: true` selects the Conformer module, anything else the pretrained-backbone module (reference main.py:69-81)
    module_cls = LidSuperviseModule if cfg.get("supervised", True) else LidModule
    module = module_cls(**cfg["module"], **cfg["model"], lang2vocab=lang2vocab, lang2index_dict=lang2index,
                        tokenizer_dict=tokenizers)
    feature = dict(data["feature"])

    def dataset(split, train):
        if data["source"] == "synthetic":
            syn = dict(data.get("synthetic") or {})
            n = syn.pop(f"{split}_items_per_lang", syn.pop("items_per_lang", 64))
            return SyntheticMergedDataset(train, lang2index, lang2vocab, items_per_lang=n, lang2tokenizer=tokenizers,
                                          seed=syn.pop("seed", 1234) + {"train": 0, "val": 1, "test": 2}[split],
                                          **{k: v for k, v in syn.items() if not k.endswith("_items_per_lang")}, **feature)
        return MergedDataset(train=train, manifest_files=manifests[split], lang2index_dict=lang2index, lang2tokenizer=tokenizers,
                             max_duration=data["max_duration"] if train else 16.7, source=data["source"], **feature)

    sets = {s: dataset(s, s == "train") for s in ("train", "val", "test")}
    sc = data["sampler_common"]
    params = dict(data["dataloader_params"])
    window = int(sc.get("bucket_window", 0))                              # > 0: batches of similar length (bucketed padding)
    lengths = getattr(sets["train"], "n_samples_of", None) if window > 0 else None
    params["train_batch_sampler"] = MutiBatchSampler(sets["train"].samplers, sc["train_batch_size"], True, rank, world,
                                                     seed=0 if world > 1 else None, lengths=lengths, bucket_window=window)
    params["val_batch_sampler"] = MutiBatchSampler(sets["val"].samplers, sc["val_batch_size"], False, rank, world, seed=1)
    params["test_batch_sampler"] = MutiBatchSampler(sets["test"].samplers, sc["test_batch_size"], False, 0, 1, seed=2)
    return module, sets, params


@hydra_main(config_path="conf", config_name="synthetic_cfg1")
def main(cfg) -> None:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    seed_everything(0)
    tcfg = dict(cfg["trainer"])
    if "LOCAL_RANK" in os.environ and tcfg.get("ddp"):                     # launched by torchrun: one process per GPU
        tcfg.update(local_rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        if tcfg.get("gpu_id") is not None:
            tcfg["gpu_id"] = int(os.environ["LOCAL_RANK"])
    module, sets, params = build(cfg, max(tcfg.get("local_rank", 0), 0), tcfg.get("world_size", 1) if tcfg.get("ddp") else 1)
    callbacks = [CkptCallback(file_name_metric=["epoch", "val_loss"], save_topk=2), LrCallback(), ProfileCallback()]
    comet = (cfg.get("logger") or {}).get("comet")                       # reference main.py:38,131: CometLogger(**comet_conf)
    loggers = [CometLogger(**comet) if comet else JsonlLogger("metrics.jsonl")]
    trainer = Trainer(callbacks=callbacks, loggers=loggers, **tcfg)
    if cfg["stage"] == "train":
        trainer.fit(module, train_dataset=sets["train"], val_dataset=sets["val"], test_dataset=sets["test"],
                    dataloader_params=params)
    else:
        trainer.test(ccml_module=module, dataloader_params=params, dataset=sets["test"])


if __name__ == "__main__":
    main()
