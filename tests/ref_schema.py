"""Test helper: a tiny on-disk corpus in the reference's ``xf_asr`` layout (``<lang>/<split>.label`` lines ``name<TAB>text``,
PCM wavs under ``<lang>/wav/train/``, one ``<lang>-vocab.txt`` per language) and a YAML written in the reference's conf schema
(the keys of lid/conf/xf_asr_supervised.yaml: ``defaults``, ``model`` with the ``model_name`` anchor + interpolation,
``supervised``, ``module``, ``data.langs[*].vocab`` as a FILE PATH, ``trainer.backend: 'nccl'``, ``logger.wandb/comet``).
Values are this repository's own; nothing is copied from the reference's files."""
import math
import os
import wave

import numpy as np

LANGS = ("Persian", "Swahili", "Vietnamese")          # the three language names the reference's xf confs use


def _write_wav(path, samples):
    pcm = np.clip(samples * 32767.0, -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(16000)
        f.writeframes(pcm.tobytes())


def make_corpus(root, n_train=8, n_val=3, seconds=(0.6, 1.0), vocab_size=12, seed=0):
    """-> {lang: {"train": label path, "val": label path, "vocab": vocab path}}"""
    rng = np.random.RandomState(seed)
    out = {}
    for k, lang in enumerate(LANGS):
        base = os.path.join(root, "xf", "data", lang)
        os.makedirs(os.path.join(base, "wav", "train"), exist_ok=True)
        symbols = [chr(ord("a") + i) for i in range(vocab_size)]
        vocab = os.path.join(root, "xf", "data", f"{lang}-vocab.txt")
        with open(vocab, "w") as f:
            f.write("\n".join(symbols) + "\n")
        paths = {"vocab": vocab}
        for split, n in (("train", n_train), ("val", n_val)):
            label = os.path.join(base, f"{split}.label")
            with open(label, "w") as f:
                for i in range(n):
                    dur = rng.uniform(*seconds)
                    t = np.arange(int(dur * 16000)) / 16000.0
                    x = 0.05 * rng.randn(t.size) + 0.2 * np.sin(2 * math.pi * (300 + 500 * k) * t)
                    name = f"{split}_{i:03d}.wav"
                    _write_wav(os.path.join(base, "wav", "train", name), x)
                    text = "".join(symbols[j] for j in rng.randint(0, vocab_size, rng.randint(3, 7)))
                    f.write(f"{name}\t{text}\n")
            paths[split] = label
        out[lang] = paths
    return out


def write_yaml(conf_dir, corpus, name="xf_like", supervised=True, gpu_id="null", total_epoch=1, extra_model=""):
    os.makedirs(os.path.join(conf_dir, "base"), exist_ok=True)
    with open(os.path.join(conf_dir, "base", "custom.yaml"), "w") as f:          # the hydra logging group the confs include
        f.write("run:\n  dir: outputs/${now:%Y-%m-%d}/${now:%H-%M}-${model.model_name}\n")
    langs = ""
    for k, lang in enumerate(LANGS):
        p = corpus[lang]
        langs += (f"    -\n      train_manifest: {p['train']}\n      val_manifest: {p['val']}\n"
                  f"      test_manifest: {p['val']}\n      lang: {lang}\n      id: {k}\n      vocab: {p['vocab']}\n")
    text = f"""defaults:
  - base: custom

model:
  model_name: &model_name >-
    lr_${{module.optimizer_param.lr}}_dr_${{model.dropout}}_bs_${{data.sampler_common.train_batch_size}}_conform_${{model.conformer_linear}}

  conformer_pure: true
  num_layers: 1
  hidden_dim: 32
  use_cer: true
  conformer_linear: true
  dropout: 0.1
  linear_dim: 64
  n_blocks: 2
  n_mels: 80
  encoder_dim: 64
  dim_head: 16
  last_dim_head: 8
  heads: 4
  ff_mult: 4
  conv_expansion_factor: 2
  conv_kernel_size: 31
  attn_dropout: 0.0
  ff_dropout: 0.0
  conv_dropout: 0.0
  double_swish: false
  sub_sampling: 2
{extra_model}
supervised: {'true' if supervised else 'false'}

module:
  optimizer_name: novograd
  optimizer_param:
    weight_decay: 0.00001
    lr: 0.01
  scheduler: tristage
  scheduler_param:
    mode: min
    factor: 0.1
    patience: 5
    cooldown: 3
    min_lr: 0.0001
  interval: 50
  freeze_tranformer_epoch: -1
  freeze_encoder_epoch: -1
  froze_wav2vec_model_epoch: -1

data:
  source: xf_asr
  feature:
    type: mel
    pad: 16
    win_length: 0.025
    t_mask: 0.05
    f_mask: 12
    mask_times: 1
    speed_shift: false
    pitch_shift: false
    reverb: false
  dataloader_params:
    pin_memory: true
    num_workers: 0
    prefetch_factor: 20
    train_batch_sampler: null
    val_batch_sampler: null
    test_batch_sampler: null
  langs:
{langs}
  sampler_common:
    train_batch_size: 4
    val_batch_size: 2
    test_batch_size: 1
  max_duration: 12

trainer:
  total_epoch: {total_epoch}
  gpu_id: {gpu_id}
  local_rank: 0
  world_size: 1
  ddp: false
  backend: 'nccl'
  init_method: env://
  accumulate_grad: 2
  master_addr: localhost
  master_port: 11488
  use_amp: false
  use_swa: false
  eval_interval: 1
  train_data_factor: 1
  log_interval: 10
  checkpoint_path: null
  resume_train_states: True

logger:
  wandb:
    project: lid_test
    entity: nobody
    name: *model_name
    wandb_id: null
  comet:
    api_key: none
    project: lid_test
    entity: nobody
    name: *model_name

stage: train
"""
    path = os.path.join(conf_dir, name + ".yaml")
    with open(path, "w") as f:
        f.write(text)
    return path
