#!/usr/bin/env python3
"""Headline benchmark: Conformer-LID training throughput in audio-seconds/sec + validation Cavg (BASELINE.json metric).

    python bench.py                                   # N = 1, 200 timed steps after 20 warm-up steps
    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the whole hot path over one batch of synthetic input already resident in HBM:
raw 3 s @ 16 kHz waveforms -> normalize + dither/pre-emphasis -> log-mel + SpecAugment (HIP) -> 12-layer d256 Conformer +
one 14-language CTC head, forward and backward (HIP, bf16 MFMA GEMMs) -> [N>1: gradient all-reduce over RCCL (bf16 payload),
SyncBN statistics] -> global-norm clip(20) + Novograd + bf16 weight refresh (HIP) -> TriStage LR step.  Batch 64 per GPU (weak
scaling), BASELINE.json configs[1]; nothing is skipped or cached between steps.  Stochastic depth is OFF in the headline
number (every layer runs every step, matching the 21.67 GFLOP/utterance accounting of SURVEY 8d); pass
--stochastic-depth to time the reference default (p=0.7, ~16 % fewer executed blocks on average).  The run exits non-zero
if any timed step produced a non-finite loss.

Prints ONE JSON line (rank 0) with the contract's fields (``value`` = K steps of whole-job audio over the wall time between
two barrier + synchronize brackets, max over ranks) plus:
  chunks_ms_per_step / median_ms_per_step : the K steps split into 5 consecutive chunks by HIP events (no extra syncs)
  roofline      : the dominant kernel (gemm_nt: every Linear / 1x1 conv forward and data gradient): algorithmic bytes of every
                  launch of one training step / their summed durations, each launch bracketed by HIP events on its own stream
                  (HBM roofline); MFMA rate of the same launches, the weight-gradient kernel, and ``features`` = the feature
                  path (normalize, dither/pre-emphasis, STFT/mel/dB/SpecAugment: 4 L + 4*80 F bytes per utterance) likewise.
                  ``traffic`` comes from a committed rocprofv3 --pmc pass of this command (``traffic_source`` names the file);
                  it is not measured by this process
  val_cavg      : (N = 1) the model keeps training on the learnable synthetic corpus (peak LR held until the CTC loss leaves its
                  plateau, then --cavg-decay steps of decay; at most --cavg-steps), then the held-out set is scored through
                  the module's own val_loop (all 14 heads, CTC-confidence LID score, lid/eer.py::CAvg)
  fit           : (N = 1) the same training through Trainer.fit INCLUDING DataLoader workers, collate, pinned H2D copies
                  (what a user of the launcher gets from in-memory data); never used for ``value``
  cpu_baseline  : the CPU oracle (oracle/, torch fp32) running the same step on all of this host's cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0               # HBM3E peak, same guide ("HBM": 8 TB/s spec, ~6.3 TB/s achievable)
SECONDS = 3.0
BATCH = 64
N_LANGS = 14
SETUP_ROUNDS = 3                    # visits of every language before warm-up (eager, capture, replay of each head's graphs)
PMC_FILE = os.path.join("profiles", "r04", "pmc_traffic.json")
PMC_FALLBACK = os.path.join("profiles", "r03", "pmc_traffic.json")
ROCPROF_STATS = (os.path.join("profiles", "r04", "bench_kernel_stats.csv"), os.path.join("profiles", "r03", "bench_kernel_stats.csv"))
FLOP_PER_UTT_TRAIN = 21.67e9       # 7.222 GFLOP forward per 3 s utterance x 3 (one head, stochastic depth off)


def build(args, rank, world, device):
    from ccml import seed_everything
    from ccml.trainer import Trainer
    from lid import hydra_lite
    import lid.main as launcher
    seed_everything(0)
    per_lang = args.batch * world * args.resident
    ov = [f"trainer.gpu_id={device.index}", "trainer.use_amp=true", f"trainer.ddp={'true' if world > 1 else 'false'}",
          f"trainer.world_size={world}", f"trainer.local_rank={rank}",
          f"trainer.backend={'gloo' if os.environ.get('LIDK_BENCH_ONE_GPU', '0') == '1' else 'nccl'}", "trainer.total_epoch=1000",
          f"data.sampler_common.train_batch_size={args.batch}", f"data.synthetic.items_per_lang={per_lang}",
          f"data.synthetic.val_items_per_lang={args.val_items}", "data.synthetic.test_items_per_lang=2",
          "module.interval=1000000", "trainer.log_interval=1000000", f"model.n_blocks={args.blocks}",
          f"module.optimizer_param.lr={args.lr}"]
    global N_LANGS
    if args.model in ("wavlm", "w2v2", "xlsr"):
        key = "wavlm_cfg" if args.model == "wavlm" else "wav2vec_cfg"
        if args.model == "xlsr" and args.blocks == 12:
            args.blocks = 24                         # XLS-R 300M's own depth unless --blocks says otherwise
        ov = [o for o in ov if not o.startswith(("model.n_blocks", "trainer.total_epoch", "module.optimizer_param.lr"))] + ["trainer.total_epoch=1000",
              f"model.{key}.encoder_layers={args.blocks}",
              f"+module.train_input_norm={'false' if args.wavlm_regime == 'heads' else 'true'}"]
        if not args.ragged:
            ov += ["data.synthetic.seconds=3.0", "data.synthetic.min_seconds=null", "data.synthetic.bucket_seconds=null"] \
                if args.model in ("w2v2", "xlsr") else ["data.synthetic.seconds=3.0"]
    if args.ragged and args.model not in ("w2v2", "xlsr"):       # SURVEY 8d's cfg5 recipe: U[1, 10] s in 1 s bins, batches of similar length
        ov += ["data.synthetic.seconds=10.0", "+data.synthetic.min_seconds=1.0", "+data.synthetic.bucket_seconds=1.0"]
    cfg = hydra_lite.load_config(os.path.join(ROOT, "speech-lid_amd", "lid", "conf"),
                                 {"conformer": "synthetic_cfg2", "wavlm": "synthetic_wavlm", "w2v2": "synthetic_w2v2", "xlsr": "synthetic_xlsr"}[args.model], ov)
    N_LANGS = len(cfg["data"]["langs"])
    module, sets, params = launcher.build(cfg, rank, world)
    module.model.use_stochastic_depth = bool(args.stochastic_depth)
    if args.model in ("wavlm", "w2v2", "xlsr") and args.wavlm_regime == "finetune":
        module.model.unfreeze_tranformer_encoder()
    trainer = Trainer(callbacks=[], loggers=[], **dict(cfg["trainer"]))
    trainer.ccml_module = module
    trainer.train_dataset, trainer.val_dataset, trainer.test_dataset = sets["train"], sets["val"], sets["test"]
    trainer.dataloader_params = params
    module.point_trainer(trainer)
    trainer.trainer_prepare()
    # the TriStage schedule spans exactly the optimizer steps this run takes (set-up, warm-up, timed, instrumented, Cavg phase)
    trainer.total_steps = SETUP_ROUNDS * N_LANGS + args.warmup + args.steps + 2 + max(args.cavg_steps, 0)
    trainer.optimizer, trainer.lr_scheduler, trainer.scheduler_param = module.config_optim()
    trainer._zero_grad()
    return cfg, module, trainer, sets


class CachedDataset(torch.utils.data.Dataset):
    """The synthetic corpus materialised once in host memory (items are pure functions of the seed): what remains per batch
    is indexing, collate (padding, SpecAugment span draws), pinning and the H2D copy - the DataLoader path of a real corpus
    held in memory."""

    def __init__(self, ds):
        self.ds = ds
        self.items = [ds[i] for i in range(len(ds))]
        self.samplers, self.train, self.type = ds.samplers, ds.train, ds.type

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def collate_fn(self, batch):
        return self.ds.collate_fn(batch)


def resident_batches(ds, rank, world, device, batch, resident, ragged=False):
    """``resident`` batches per language, built once and moved to HBM (each rank gets different utterances of the language).
    Order: round-robin over languages, so consecutive steps train different heads.  ragged: a language's utterances are sorted by
    length and cut into batches (bucketed padding, what MutiBatchSampler(bucket_window) yields); each batch carries the seconds
    of TRUE audio it holds (``_audio_s``), which is what the throughput counts - padding is not audio."""
    per_lang = len(ds) // N_LANGS
    out = []
    for r in range(resident):
        for k in range(N_LANGS):
            base = k * per_lang
            order = list(range(per_lang))
            if ragged:
                order.sort(key=lambda i: ds.n_samples_of(base + i))
            idx = [base + order[((r * world + rank) * batch + j) % per_lang] for j in range(batch)]
            b = list(ds.collate_fn([ds[i] for i in idx]))
            b.append(sum(ds.n_samples_of(i) for i in idx) / 16000.0 if ragged else batch * SECONDS)
            b[0] = [w.to(device) for w in b[0]] if isinstance(b[0], list) else b[0].to(device)
            for j in (1, 2, 3, 5):
                host = b[j]
                b[j] = host.to(device)
                if not host.is_floating_point():          # as Trainer.batch_to_device does: small integer metadata (the batch's
                    b[j]._host = host                     # language id) stays readable on the host without a device sync

            out.append(b)
    return out


def log(msg):
    """Progress to stderr (the JSON line is the only thing on stdout); also keeps a long run visibly alive."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_info():
    """(CPU model string, cores this process may actually use) - the smaller of the scheduler affinity and the cgroup CPU
    quota (a container on a shared host sees every core in the affinity mask but is throttled to its quota)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                usable = max(1, min(usable, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return model, usable


def cpu_baseline(module, ds, steps, warm, batch, threads, cpu_model, budget_s=40.0):
    """The oracle (torch-CPU fp32 restatement, pinned to the reference) running the same training step on host cores.
    Bounded: stops early (never before 3 timed steps) once ``budget_s`` seconds of CPU work have been spent."""
    from oracle import conformer as oc
    from oracle import features as of
    from oracle import optim as oo
    torch.set_num_threads(threads)
    cfg = module.model.cfg
    ocfg = oc.ModelCfg(lang2vocab=cfg.lang2vocab, lang2index=cfg.lang2index, n_blocks=cfg.n_blocks, encoder_dim=cfg.d,
                       dim_head=cfg.dim_head, heads=cfg.heads, last_dim_head=cfg.last_dim_head, dropout=cfg.dropout)
    sd = {k: v.detach().cpu().clone() for k, v in module.model.state_dict().items()}
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    states = {k: oo.NovogradState() for k in names}
    per_lang = len(ds) // N_LANGS
    times = []
    began = time.time()
    for step in range(steps + warm):
        if len(times) >= 3 and time.time() - began > budget_s:
            break
        k = step % N_LANGS
        lang = list(cfg.lang2vocab)[k]
        items = [ds[k * per_lang + j % per_lang] for j in range(batch)]
        t0 = time.time()
        wav = of.normalize_wav(torch.stack([it[0] for it in items]))
        wav = of.dither_preemphasis(wav, torch.rand_like(wav))
        mel = of.wav2mel(wav, pad=16)
        mel = torch.stack([of.apply_specaug(m, of.draw_specaug_spans(m.shape[-1], 80, 0.05, 12, 1)) for m in mel])
        feats = mel.transpose(1, 2).contiguous()
        p = {n: sd[n].requires_grad_(True) for n in names}
        full = {**sd, **p}
        opts = oc.RunOpts(training=True, keep_layers=[True] * cfg.n_blocks)
        out, _ = oc.forward(feats, full, ocfg, lang, opts)
        texts = torch.stack([it[1] for it in items])
        loss = oc.ctc_loss(out[lang], texts, torch.ones(batch), torch.ones(batch), blank=cfg.lang2vocab[lang])
        loss.backward()
        with torch.no_grad():
            act = [n for n in names if sd[n].grad is not None]
            oo.clip_grad_norm([sd[n].grad for n in act], 20.0)
            oo.novograd_step([sd[n] for n in act], [sd[n].grad for n in act], [states[n] for n in act], lr=1e-3,
                             weight_decay=1e-5)
            for n in names:
                sd[n].grad = None
                sd[n].requires_grad_(False)
            sd.update(opts.bn_buffers)
        if step >= warm:
            times.append(time.time() - t0)
        log(f"cpu baseline step {step + 1}/{steps + warm}: {time.time() - t0:.2f} s")
    steps = len(times)
    times.sort()
    t = times[len(times) // 2]
    return {"value": round(batch * SECONDS / t, 2), "unit": "audio-seconds/sec", "cores": threads, "cpu_model": cpu_model,
            "kind": "port",
            "sample": f"median of {steps} steps of batch {batch} (3 s utterances) after {warm} warm-up steps, {t:.2f} s/step, "
                      f"fp32 torch-CPU oracle with torch.set_num_threads({threads}) = every core this process may use"}


def cpu_baseline_wavlm(module, ds, steps, warm, batch, threads, cpu_model):
    """The CPU oracle (oracle/wavlm.py + oracle/conformer.py, pinned to the reference) running the frozen-backbone training step:
    waveform normalisation, backbone forward without gradients, head forward + backward under autograd, Adam on the head."""
    from oracle import conformer as oc
    from oracle import features as of
    from oracle import wavlm as ow
    torch.set_num_threads(threads)
    cfg, wcfg = module.model.cfg, module.model.backbone.cfg
    hcfg = oc.ModelCfg(lang2vocab=cfg.lang2vocab, lang2index=cfg.lang2index, n_blocks=0, encoder_dim=cfg.d,
                       last_dim_head=cfg.last_dim_head, last_heads=cfg.last_heads, dropout=0.0, hidden_dim=cfg.hidden_dim)
    sd = {k: v.detach().cpu().clone() for k, v in module.model.state_dict().items()}
    bb = {k[len("model.featurizer.model."):]: v for k, v in sd.items() if k.startswith("model.featurizer.model.")}
    heads = {k: v for k, v in sd.items() if not k.startswith("model.featurizer.")}
    names = [k for k, v in heads.items() if v.is_floating_point() and "running_" not in k]
    per_lang = len(ds) // N_LANGS
    times = []
    for step in range(steps + warm):
        k = step % N_LANGS
        lang = list(cfg.lang2vocab)[k]
        items = [ds[k * per_lang + j % per_lang] for j in range(batch)]
        t0 = time.time()
        wav = of.normalize_wav(torch.stack([it[0] for it in items]))
        with torch.no_grad():
            feat = ow.backbone(wav, bb, wcfg)
        p = {n: heads[n].requires_grad_(True) for n in names if f".{lang}." in n}
        opts = oc.RunOpts(training=True)
        logits = oc.head(feat, {**heads, **p}, hcfg, lang, opts)
        texts = torch.stack([it[1] for it in items])
        loss = oc.ctc_loss(logits, texts, torch.ones(batch), torch.ones(batch), blank=cfg.lang2vocab[lang])
        loss.backward()
        with torch.no_grad():
            for n, t in p.items():                            # plain SGD stands in for Adam's few element-wise passes
                t -= 1e-4 * t.grad
                t.grad = None
                t.requires_grad_(False)
            heads.update(opts.bn_buffers)
        if step >= warm:
            times.append(time.time() - t0)
        log(f"cpu baseline (wavlm) step {step + 1}/{steps + warm}: {time.time() - t0:.2f} s")
    times.sort()
    t = times[len(times) // 2]
    return {"value": round(batch * SECONDS / t, 2), "unit": "audio-seconds/sec", "cores": threads, "cpu_model": cpu_model,
            "kind": "port",
            "sample": f"median of {len(times)} steps of batch {batch} (3 s utterances) after {warm} warm-up, {t:.2f} s/step, fp32 "
                      f"torch-CPU oracle (backbone forward without gradients + head forward/backward), {threads} threads"}


def _pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (tools/gpu_pmc_bench.sh on this same
    command): FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE counts 64 B per 128-B request on gfx950 and is doubled, as
    MI355X_MICROARCH.md (HBM) prescribes; WRITE_SIZE is exact for 16-byte stores.  -> (bytes, source file) or (None, None)."""
    for rel in (PMC_FILE, PMC_FALLBACK):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        raw = json.load(open(path))
        hits = [v for name, v in raw.items() if kernel_prefix in name and "FETCH_SIZE_KB_per_launch_raw" in v
                and "WRITE_SIZE_KB_per_launch_raw" in v]
        n = sum(v.get("launches_fetch", 0) for v in hits)
        if not n:
            continue
        kb = sum((2.0 * v["FETCH_SIZE_KB_per_launch_raw"] + v["WRITE_SIZE_KB_per_launch_raw"]) * v["launches_fetch"] for v in hits)
        return round(kb / n * 1024.0), rel                              # launch-weighted mean over the kernel's instantiations
    return None, None


def _rocprof_avg_us(prefixes):
    """Launch-weighted average duration of the kernels whose name contains one of ``prefixes`` in the committed rocprofv3
    --kernel-trace --stats summary of this command (profiles/r04/bench_kernel_stats.csv) -> (us, launches, file) or Nones."""
    import csv
    for rel in ROCPROF_STATS:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        tot, n = 0.0, 0
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if any(px in row["Name"] for px in prefixes):
                    tot += float(row["TotalDurationNs"])
                    n += int(row["Calls"])
        if n:
            return round(tot / n / 1e3, 3), n, rel
    return None, None, None


def _replay_ms(calls, reps=9):
    """Device time of ``calls`` (a list of zero-argument launchers) issued back-to-back from ONE captured hipGraph, median of
    ``reps`` replays bracketed by two HIP events on the replay stream.  Nothing is subtracted: the figure contains the gaps
    between consecutive kernel nodes, so it can only under-state the kernels' own rate."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for c in calls:
            c()
    g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    return sorted(times)[len(times) // 2]


def roofline(trainer, batches, step_fn):
    """The step's three big kernel families - gemm_nt (Linear / 1x1 conv forward and data gradients), the fused FeedForward
    kernels (ffn_fwd / ffn_bwd: LayerNorm + both projections + residual, and their data gradients + LayerNorm backward) and the
    grouped weight-gradient GEMM (gemm_tn_grouped) - plus the feature path, measured live and WITHOUT any calibration constant:
    one training step runs through the eager launch path with every launch of these kernels recorded (operands are the step's
    own workspace buffers); the recorded launches of a family are then captured into ONE hipGraph, back-to-back in step order,
    and that graph is replayed between two HIP events on its stream: average launch = replay time / launches.  The figure
    includes the gap between consecutive kernel nodes (it can only under-state the kernel); the launch-weighted average of the
    same kernels in the committed rocprofv3 --kernel-trace --stats summary of this command is printed beside it.
    The headline object describes the family with the most device time per step (``family_ms_per_step`` lists all three).
    Every one of them sits below the machine balance (100-160 FLOP per algorithmic byte against 2500 TF/s / 8 TB/s = 312), so
    the roofline that bounds them is HBM: achieved = algorithmic bytes (operands read once, outputs written once; DESIGN.md
    section 5) / measured time.  The MFMA rate of the same launches is reported beside it."""
    from lid.audio_processor import WaveBatch
    eng = trainer.engine
    k = eng.k
    names = ("gemm_nt", "gemm_nt_bn_sums", "gemm_tn", "ffn_fwd", "ffn_bwd", "build_tn_group", "gemm_tn_grouped")
    orig = {n: getattr(k, n) for n in names if hasattr(k, n)}
    orig_mel = WaveBatch._compute_mel
    rec = {"nt": [], "tn": [], "ffn": [], "tng": [], "feat": []}
    groups = {}

    def esz(t):
        return t.element_size()

    def timed_nt(A, B, out, *a, M=None, N=None, K=None, **kw):
        m, kk, n = (A.shape[0] if M is None else M), (A.shape[1] if K is None else K), (B.shape[0] if N is None else N)
        a_cols = min(kk, A.stride(0))                 # a strided-view convolution operand: each input value is read once
        nbytes = m * a_cols * esz(A) + n * kk * esz(B) + m * n * esz(out)
        for key in ("out2", "aux", "res"):
            if kw.get(key) is not None:
                nbytes += m * n * esz(kw[key])
        if kw.get("bias") is not None:
            nbytes += n * 4
        rec["nt"].append((lambda: orig["gemm_nt"](A, B, out, *a, M=M, N=N, K=K, **kw), 2.0 * m * n * kk, nbytes))
        return orig["gemm_nt"](A, B, out, *a, M=M, N=N, K=K, **kw)

    def timed_nt_bn(A, B, out, c, mean, rstd, gamma, beta, partial, M=None, N=None, K=None):
        # the conv module's pw2 data gradient with the BatchNorm backward sums in its epilogue: + the BatchNorm input c read once
        m, kk, n = (A.shape[0] if M is None else M), (A.shape[1] if K is None else K), (B.shape[0] if N is None else N)
        nbytes = m * kk * esz(A) + n * kk * esz(B) + m * n * esz(out) + m * n * esz(c) + (m // 64) * 2 * 2 * n * 4
        call = lambda: orig["gemm_nt_bn_sums"](A, B, out, c, mean, rstd, gamma, beta, partial, M=M, N=N, K=K)
        r = call()
        if r:
            rec["nt"].append((call, 2.0 * m * n * kk, nbytes))
        return r

    def timed_tn(X, Y, C, *a, M=None, N1=None, N2=None, **kw):
        m, n1, n2 = (X.shape[0] if M is None else M), (X.shape[1] if N1 is None else N1), (Y.shape[1] if N2 is None else N2)
        nbytes = m * n1 * esz(X) + m * n2 * esz(Y) + n1 * n2 * 4
        rec["tn"].append((lambda: orig["gemm_tn"](X, Y, C, *a, M=M, N1=N1, N2=N2, **kw), 2.0 * m * n1 * n2, nbytes))
        return orig["gemm_tn"](X, Y, C, *a, M=M, N1=N1, N2=N2, **kw)

    def timed_ffn_fwd(x, W1, b1, W2, b2, xo, **kw):
        m, d, ff = x.shape[0], x.shape[1], W1.shape[0]
        # x (f32) in, xo (f32) out, W1 + W2 + biases, a and u out (T), h out - or in, when the LayerNorm pair produced it
        nbytes = 2 * m * d * 4 + 2 * ff * d * 2 + (ff + d) * 4 + 2 * m * ff * 2 + m * d * 2 + (0 if kw.get("h_in") is not None else 8 * m)
        nl = kw.get("next_ln") or {}                  # the consuming LayerNorm(s) applied in the epilogue: their outputs + statistics
        nbytes += sum(m * d * t.element_size() for key, t in nl.items() if key in ("yA32", "yAT", "yBT") and t is not None)
        nbytes += 8 * m * (("gA" in nl) + (nl.get("gB") is not None))
        rec["ffn"].append((lambda: orig["ffn_fwd"](x, W1, b1, W2, b2, xo, **kw), 4.0 * m * d * ff, nbytes))
        return orig["ffn_fwd"](x, W1, b1, W2, b2, xo, **kw)

    def timed_ffn_bwd(dyT, a, W1T, W2T, da, **kw):
        m, d, ff = dyT.shape[0], dyT.shape[1], a.shape[1]
        nbytes = m * d * 2 + 2 * m * ff * 2 + 2 * ff * d * 2            # dyT, a in, da out, both weight operands
        if kw.get("dh") is not None:
            nbytes += m * d * 2
        else:                                                          # x + dres in, dx (+ dxT) out, statistics, partial rows
            nbytes += m * d * 4 * (2 if kw.get("dres") is not None else 1) + (m * d * 4 if kw.get("dx") is not None else 0) \
                + (m * d * 2 if kw.get("dxT") is not None else 0) + 8 * m
        rec["ffn"].append((lambda: orig["ffn_bwd"](dyT, a, W1T, W2T, da, **kw), 4.0 * m * d * ff, nbytes))
        return orig["ffn_bwd"](dyT, a, W1T, W2T, da, **kw)

    def timed_build(entries, **kw):
        grp = orig["build_tn_group"](entries, **kw)
        groups[id(grp[0])] = (sum(2.0 * M * n1 * n2 for _, _, _, _, M, n1, n2 in entries),
                              sum(M * n1 * 2 + M * n2 * 2 + n1 * n2 * 4 for _, _, _, _, M, n1, n2 in entries))
        return grp

    def timed_tng(grp):
        if id(grp[0]) in groups:
            fl, by = groups[id(grp[0])]
            rec["tng"].append((lambda: orig["gemm_tn_grouped"](grp), fl, by))
        return orig["gemm_tn_grouped"](grp)

    def timed_mel(self):
        Bn, Ln = self.wav.shape
        F_ = 1 + (Ln + 2 * self.pad) // 160
        rec["feat"].append((lambda: orig_mel(self), 0.0, Bn * (4 * Ln + 4 * self.n_mels * F_)))
        return orig_mel(self)

    graphs_on = eng.graphs.enabled
    eng.graphs.enabled = False
    if hasattr(batches[0][0], "_mel"):
        batches[0][0]._mel = batches[0][0]._mel_ready = None    # drop a prefetched copy: this step computes its own features
    # The grouped launches read their shapes from descriptor tables the engine caches per workspace.  Set the caches aside (NOT
    # freed: captured hipGraphs hold raw pointers into those tables) so that this step rebuilds its tables under the recorder,
    # and put them back afterwards.
    held = [(w_, w_.__dict__.pop("_tn_groups")) for w_ in list(eng._work.values()) if "_tn_groups" in w_.__dict__]
    wraps = {"gemm_nt": timed_nt, "gemm_nt_bn_sums": timed_nt_bn, "gemm_tn": timed_tn, "ffn_fwd": timed_ffn_fwd, "ffn_bwd": timed_ffn_bwd,
             "build_tn_group": timed_build, "gemm_tn_grouped": timed_tng}
    for n in orig:
        setattr(k, n, wraps[n])
    WaveBatch._compute_mel = timed_mel
    try:
        step_fn(0, batches[0])
        torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(k, n, f)
        WaveBatch._compute_mel = orig_mel
        eng.graphs.enabled = graphs_on
    fresh = [(w_, w_.__dict__.get("_tn_groups")) for w_, _ in held]      # tables built by this step: referenced by the recorder
    for w_, cache in held:
        w_.__dict__["_tn_groups"] = cache

    def tot(kind):
        if not rec[kind]:
            return 0.0, 0.0, 0, 0
        ms = _replay_ms([c for c, _, _ in rec[kind]])
        return ms, sum(f for _, f, _ in rec[kind]), sum(n for _, _, n in rec[kind]), len(rec[kind])

    fam = {kind: tot(kind) for kind in ("nt", "ffn", "tng", "tn", "feat")}
    trainer._zero_grad()                               # the replayed weight-gradient launches accumulated into the arena
    torch.cuda.synchronize()
    del fresh
    method = ("all launches of one step captured back-to-back into one hipGraph, replay time / launches between two HIP events "
              "(nothing subtracted; contains the inter-node gaps)")
    label = {"nt": ("gemm_nt (gemm_nt_bf16_pipe_kernel + gemm_nt_bf16_direct_kernel, 64x64 tiles)", ("gemm_nt_bf16_",)),
             "ffn": ("fused FeedForward (ffn_fwd_kernel + ffn_bwd_kernel, csrc/ffn.hip: 48-row workgroups at this M, weights by LDS-DMA)",
                     # (the GEMM-only instantiations ffn_bwd_kernel<*, true, *> are lidk_dgrad_ln_bwd launches: not this family)
                     ("ffn_fwd_kernel", "ffn_bwd_kernel<true, false,", "ffn_bwd_kernel<false, false,")),
             "tng": ("grouped weight-gradient GEMM (gemm_tn_grouped_dma_kernel: 128x128 tiles, 2 row chunks, operands by LDS-DMA ring)", ("gemm_tn_grouped",)),
             "tn": ("gemm_tn_bf16_kernel<64,64>", ("gemm_tn_bf16_kernel",))}

    def describe(kind):
        ms, fl, by, n = fam[kind]
        if not n or ms <= 0:
            return None
        gbs, tfs = by / (ms * 1e-3) / 1e9, fl / (ms * 1e-3) / 1e12
        rp_us, rp_n, rp_src = _rocprof_avg_us(label[kind][1])
        traffic, src = _pmc_traffic(label[kind][1][0]) if len(label[kind][1]) == 1 else (None, None)
        if len(label[kind][1]) > 1:                    # launch-weighted over the family's kernels
            parts = [(_pmc_traffic(px), _rocprof_avg_us((px,))) for px in label[kind][1]]
            if all(t[0] is not None and r[1] for t, r in parts):
                traffic = round(sum(t[0] * r[1] for t, r in parts) / sum(r[1] for _, r in parts))
                src = parts[0][0][1]
        mfma_bound = fl / max(by, 1) > MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        d = {"bound": "mfma" if mfma_bound else "hbm",
             "achieved": round(tfs if mfma_bound else gbs, 1), "peak": MFMA_BF16_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
             "unit": "TFLOP/s" if mfma_bound else "GB/s",
             "frac": round(tfs / MFMA_BF16_PEAK_TFLOPS if mfma_bound else gbs / HBM_PEAK_GBS, 4),
             "traffic": traffic,
             "traffic_source": (f"{src} (committed rocprofv3 --pmc pass of this command; not measured in this run)" if src else None),
             "kernel": label[kind][0], "launches_per_step": n, "avg_launch_us": round(ms * 1e3 / n, 2),
             "kernel_ms_per_step": round(ms, 3), "algorithmic_bytes_per_launch": round(by / n), "flops_per_launch": round(fl / n),
             "flop_per_byte": round(fl / max(by, 1), 1), "method": method,
             "rocprof_avg_us": rp_us, "rocprof_launches": rp_n, "rocprof_source": rp_src,
             "frac_from_rocprof": (round(by / n / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rp_us and not mfma_bound else None),
             "mfma": {"achieved": round(tfs, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfs / MFMA_BF16_PEAK_TFLOPS, 4)},
             "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}}
        return d

    big = max(("nt", "ffn", "tng"), key=lambda kind: fam[kind][0])
    out = describe(big)
    out["family_ms_per_step"] = {label[kind][0].split(" (")[0]: round(fam[kind][0], 3) for kind in ("nt", "ffn", "tng") if fam[kind][3]}
    for kind, key in (("nt", "gemm_nt"), ("ffn", "feedforward_fused"), ("tng", "wgrad_grouped"), ("tn", "wgrad_kernel")):
        if kind != big:
            out[key] = describe(kind)
    ms3, _, by3, n3 = fam["feat"]
    out["features"] = None
    if n3 and ms3 > 0:
        f_gbs = by3 / (ms3 * 1e-3) / 1e9
        out["features"] = {"bound": "hbm", "achieved": round(f_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(f_gbs / HBM_PEAK_GBS, 4),
                           "kernels": "waveform statistics + STFT/mel (normalise, dither and pre-emphasis in its frame load) + dB floor / SpecAugment (one batch)",
                           "ms_per_batch": round(ms3, 4), "algorithmic_bytes_per_batch": int(by3)}
    return out


def phase_times(trainer, batches, step_fn):
    """Device time of the phases of ONE more step (graphs on, as in the timed loop): features + forward + CTC, backward,
    clip + optimizer + weight refresh.  HIP events on the main stream around the module's train_loop, loss.backward and the
    trainer's optimizer step."""
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("t0", "fwd", "bwd", "opt")}
    mod = trainer.ccml_module
    orig_loop, orig_opt = mod.train_loop, trainer._optimizer_step

    def loop(batch):
        ev["t0"].record()
        out = orig_loop(batch)
        ev["fwd"].record()
        return out

    def opt():
        ev["bwd"].record()
        orig_opt()
        ev["opt"].record()

    mod.train_loop, trainer._optimizer_step = loop, opt
    try:
        step_fn(1, batches[1 % len(batches)])
        torch.cuda.synchronize()
    finally:
        mod.train_loop, trainer._optimizer_step = orig_loop, orig_opt
    return {"features_forward_loss": round(ev["t0"].elapsed_time(ev["fwd"]), 3),
            "backward": round(ev["fwd"].elapsed_time(ev["bwd"]), 3),
            "clip_optimizer_refresh": round(ev["bwd"].elapsed_time(ev["opt"]), 3)}


def cavg_phase(args, module, trainer, batches, step_fn, first_step):
    """Keep training on the learnable corpus (the timed steps were ordinary training steps of the same run), then score the
    held-out utterances through the module's own validation loop.

    The learning-rate recipe of this phase is the harness's own (the reference's TriStage is a function of a step count fixed
    in advance, and the step at which CTC leaves its "right number of tokens, random identities" plateau - loss ~ text_len *
    ln(vocab) - varies from run to run): ramp to the peak LR, HOLD it until the smoothed training loss has fallen below a
    third of that plateau, then decay exponentially to 2 % of the peak over --cavg-decay steps; --cavg-steps caps the phase."""
    import math
    t0 = time.perf_counter()
    peak, ramp = float(args.lr), 300
    groups = trainer.optimizer.param_groups
    lr0 = float(groups[0]["lr"])
    trainer.lr_scheduler = None                              # this phase sets the learning rate itself
    text_len = int(batches[0][1].shape[1]) if hasattr(batches[0][1], "shape") else 20
    vocab = max(int(v) for v in module.model.cfg.lang2vocab.values())
    plateau = text_len * math.log(max(vocab, 2))                 # loss of "right token count, uniform identities"
    recent, broke_at, last, steps_run = [], None, None, 0
    for i in range(args.cavg_steps):
        if broke_at is None:
            lr = lr0 + (peak - lr0) * min(1.0, (i + 1) / ramp)
        else:
            lr = peak * 0.02 ** min(1.0, (i - broke_at) / max(args.cavg_decay, 1))
        for g in groups:
            g["lr"] = lr
        out, loss, _ = step_fn(first_step + i, batches[i % len(batches)], batches[(i + 1) % len(batches)])
        last, steps_run = loss, i + 1
        if (i + 1) % 25 == 0:                                # one host read of the loss every 25 steps
            recent = (recent + [float(loss)])[-8:]
            if broke_at is None and len(recent) == 8 and sum(recent) / 8 < plateau / 3:
                broke_at = i
                log(f"cavg phase: plateau left at extra step {i + 1} (smoothed loss {sum(recent) / 8:.2f} < {plateau / 3:.1f}); decaying")
        if (i + 1) % 500 == 0:
            log(f"cavg phase: {i + 1} extra steps, loss {float(loss):.3f}, lr {lr:.5f}")
        if args.cavg_eval_every > 0 and (i + 1) % args.cavg_eval_every == 0 and i + 1 < args.cavg_steps:
            trainer._evaluate(0)                       # exploration aid: intermediate validation (not part of the default run)
            log(f"cavg phase: step {i + 1} validation {module.last_val}")
            trainer.model.train()
        if broke_at is not None and i - broke_at >= args.cavg_decay:
            break
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    t1 = time.perf_counter()
    trainer._evaluate(0)
    torch.cuda.synchronize()
    val = dict(module.last_val)
    log(f"cavg phase: validation {val}")
    return {"val_cavg": val["cavg"], "val_eer": round(float(val["eer"]), 4), "val_cer": round(float(val["val_wer"]), 4),
            "val_loss": round(float(val["val_loss"]), 4), "train_loss_last": round(float(last), 4) if last is not None else None,
            "optimizer_steps_total": int(trainer.current_step), "extra_train_steps": steps_run,
            "plateau_left_at_extra_step": None if broke_at is None else broke_at + 1,
            "extra_train_seconds": round(train_s, 1), "validation_seconds": round(time.perf_counter() - t1, 1),
            "held_out_utterances": len(trainer.val_dataset), "chance": 0.5,
            "note": "learnable synthetic corpus (tone-pair transcripts, language-specific symbol tables); peak LR held until the "
                    "CTC loss leaves its plateau, then decayed; CTC-confidence LID score of all 14 heads per utterance -> "
                    "score_to_prob -> lid/eer.py::CAvg"}


class _EpochTimer:
    """Callback for the Trainer.fit measurement: wall time of each training epoch (device drained at the end of the epoch)."""

    def __init__(self):
        self.trainer, self.spans, self._t = None, [], None
        self.step_stamps, self._stamps = [], []            # host time stamps of every optimizer step, per epoch

    def add_trainer(self, trainer):
        self.trainer = trainer

    def before_train_epoch(self, value):
        torch.cuda.synchronize()
        self._t = time.perf_counter()
        self._stamps = []

    def after_train_epoch(self, value):
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.spans.append(now - self._t)
        self.step_stamps.append(self._stamps + [now])

    def after_train_loop(self, value):
        self._stamps.append(time.perf_counter())

    def _noop(self, value):
        pass

    after_eval_loop = after_eval_epoch = test_loop_end = _noop


def fit_throughput(args, cfg, module, sets, params, device):
    """Trainer.fit over the cached corpus with DataLoader workers: per-epoch wall time -> audio-s/s including data loading."""
    from ccml.trainer import Trainer
    from ccml.utils.profile import _time_cost_recoder as rec
    train = CachedDataset(sets["train"])
    timer = _EpochTimer()
    tcfg = dict(cfg["trainer"])
    tcfg.update(total_epoch=args.fit_epochs, eval_interval=10 ** 6)
    trainer = Trainer(callbacks=[timer], loggers=[], **tcfg)
    p = dict(params)
    p.update(num_workers=args.fit_workers, prefetch_factor=4, pin_memory=True, persistent_workers=args.fit_workers > 0)
    rec._clear()
    trainer.fit(module, train_dataset=train, val_dataset=sets["val"], test_dataset=sets["test"], dataloader_params=p)
    n_batches = len(trainer.train_dataloader)
    steps_all = max(n_batches * args.fit_epochs, 1)
    host = {k: round(v / steps_all * 1e3, 3) for k, v in rec.values_map.items()}      # host ms per step, all epochs
    log(f"fit: epoch seconds {[round(x, 2) for x in timer.spans]}, host ms/step by section {host}")
    spans = timer.spans[1:] or timer.spans                        # the first epoch starts the workers and captures graphs
    best = sorted(spans)[len(spans) // 2]
    # inside an epoch, away from its first batches (the DataLoader re-primes its prefetch queue at every epoch start, which a
    # 56-batch epoch feels and a real one does not): from the 8th step's stamp to the drained end of the epoch
    steady = sorted((st[-1] - st[7]) / (len(st) - 8) for st in (timer.step_stamps[1:] or timer.step_stamps) if len(st) > 16)
    steady_ms = round(steady[len(steady) // 2] * 1e3, 3) if steady else None
    return {"audio_s_per_s": round(n_batches * args.batch * SECONDS / best, 1), "ms_per_step": round(best / n_batches * 1e3, 3),
            "ms_per_step_within_epoch": steady_ms,
            "batches_per_epoch": n_batches, "epochs_timed": len(spans), "num_workers": args.fit_workers,
            "epoch_seconds": [round(x, 3) for x in timer.spans], "host_ms_per_step_all_epochs": host,
            "includes": "DataLoader worker processes (in-memory corpus), collate with SpecAugment span draws, pinned-memory H2D "
                        "copy of raw waveforms, GPU features, training step; median epoch after the first "
                        "(host_ms_per_step_all_epochs averages over ALL epochs, so it carries the first epoch's one-off graph "
                        "captures and worker start-up)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["conformer", "wavlm", "w2v2", "xlsr"], default="conformer",
                    help="conformer = BASELINE configs[1] (the headline); wavlm = configs[3]: WavLM-base backbone + heads; "
                         "w2v2 = configs[4]: wav2vec2-base backbone (key padding mask, hidden-state mix) + heads")
    ap.add_argument("--ragged", action="store_true",
                    help="variable-length utterances U[1, 10] s in 1 s bins with bucketed padding (BASELINE configs[4]'s data shape); "
                         "audio-seconds count TRUE audio")
    ap.add_argument("--wavlm-regime", choices=["heads", "frozen", "finetune"], default="frozen",
                    help="heads = gradient stops at the features (train_input_norm=false: backbone = 2 graphs); frozen = the "
                         "reference's first-epoch regime (encoder frozen, layer_norm + mask_emb train: data gradients through the "
                         "transformer); finetune = transformer encoder un-frozen (after freeze_tranformer_epoch)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chunks", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--lr", type=float, default=0.003,
                    help="peak Novograd LR.  0.003: on the synthetic corpus CTC leaves its plateau reliably after ~1.2-1.4 k steps; at the "
                         "config file's 0.01 that is a coin flip and at >= 0.03 it never does (profiles/r02/cavg_runs.txt)")
    ap.add_argument("--resident", type=int, default=4, help="resident batches per language (256 utterances per language at batch 64)")
    ap.add_argument("--val-items", type=int, default=16, help="held-out utterances per language")
    ap.add_argument("--stochastic-depth", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cavg-steps", type=int, default=10000, help="cap on the extra training steps before the validation Cavg (0 = skip)")
    ap.add_argument("--cavg-decay", type=int, default=1500, help="LR decay steps once the CTC loss has left its plateau")
    ap.add_argument("--cavg-eval-every", type=int, default=0, help="also validate every K extra steps (exploration aid)")
    ap.add_argument("--fit-epochs", type=int, default=4, help="epochs of the Trainer.fit measurement (0 = skip)")
    ap.add_argument("--fit-workers", type=int, default=4)
    args = ap.parse_args()
    if args.model in ("wavlm", "w2v2", "xlsr") or args.ragged:        # secondary lines: no Cavg phase, no fit leg
        args.cavg_steps, args.fit_epochs = 0, 0

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the lidk path has no CPU fallback")
    # Rehearsal switches for a ONE-GPU box (tests only; the driver's N > 1 runs use neither): every rank on cuda:0 and gloo
    # instead of RCCL, which refuses two ranks on one device
    one_gpu = os.environ.get("LIDK_BENCH_ONE_GPU", "0") == "1"
    device = torch.device("cuda:0" if one_gpu else f"cuda:{local}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        args.cavg_steps = 0                      # the Cavg / fit / CPU legs belong to the N = 1 run (see the docstring)
        args.fit_epochs = 0

    cfg, module, trainer, sets = build(args, rank, world, device)
    ds = sets["train"]
    log("model built; materialising resident batches")
    batches = resident_batches(ds, rank, world, device, args.batch, args.resident, ragged=args.ragged)
    batch_audio = [b.pop() for b in batches]
    log(f"{len(batches)} resident batches in HBM; set-up + warm-up")
    nb = len(batches)
    n_total = 10 ** 9

    def step_fn(i, batch, upcoming=None):
        return trainer.train_step(i, batch, n_total, upcoming)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed set-up (the analogue of JIT compilation): each language head's block sequence is captured into a hipGraph on
    # its second use, so visit every language three times before the contract's W warm-up steps.  These are ordinary training
    # steps; nothing they compute is reused by the timed steps.
    it = 0
    if trainer.engine.graphs.enabled and not args.stochastic_depth:
        for _ in range(SETUP_ROUNDS * N_LANGS):
            step_fn(it, batches[it % nb])
            it += 1
    # From here on every step also starts the NEXT batch's feature kernels (normalize, dither, STFT/mel) on the feature
    # stream, as Trainer.fit does: each timed step still contains exactly one batch's worth of feature work.
    for _ in range(args.warmup):
        step_fn(it, batches[it % nb], batches[(it + 1) % nb])
        it += 1
    chunks = max(1, min(args.chunks, args.steps))
    bounds = [args.steps * c // chunks for c in range(chunks + 1)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    losses = []
    sync()
    t0 = time.perf_counter()
    marks[0].record()
    burst = min(8, args.steps)                     # host cost per step: the first steps after a sync, while the device queue is
    host_issue = 0.0                               # still short (later the host is throttled by the queue depth, not by its work)
    audio_s_rank = 0.0
    for i in range(args.steps):
        _, loss, _ = step_fn(it, batches[it % nb], batches[(it + 1) % nb])
        audio_s_rank += batch_audio[it % nb]
        it += 1
        losses.append(loss)
        if i + 1 == burst:
            host_issue = (time.perf_counter() - t0) / burst
        if i + 1 in bounds[1:]:
            marks[bounds.index(i + 1, 1)].record()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    log(f"timed {args.steps} steps: {elapsed / args.steps * 1e3:.3f} ms/step")
    loss_vals = torch.stack([l.float() for l in losses]).cpu()
    finite = bool(torch.isfinite(loss_vals).all())
    chunk_ms = [marks[c].elapsed_time(marks[c + 1]) / max(bounds[c + 1] - bounds[c], 1) for c in range(chunks)]

    # Exposed communication (N > 1): the same steps again with every data-path collective stubbed out (gradient exchange skipped,
    # SyncBatchNorm all-reduce a no-op: the ranks then train on local statistics - timing only, after the timed region).
    comm = None
    if world > 1:
        eng = trainer.engine
        keep_sync, keep_stat = trainer._sync_grads, eng.stat_allreduce
        n_probe = max(10, args.steps // 4)

        keep_ready, keep_flush, keep_buf = eng.on_stage_grads_ready, getattr(trainer, "_dp_flush", None), getattr(trainer, "_dp_buffer_hook", None)

        def probe(stub):
            nonlocal it
            if stub:                                     # the probe's stubs live HERE, not in the trainer: the hooks become no-ops
                eng.stat_allreduce = (lambda t: None) if keep_stat is not None else None
                eng.on_stage_grads_ready = (lambda stage: None) if keep_ready is not None else None
                trainer._dp_flush = (lambda: None) if keep_flush is not None else None
                if hasattr(trainer.model, "on_backbone_grads_ready"):
                    trainer.model.on_backbone_grads_ready = (lambda buf: None)
            sync()
            t_ = time.perf_counter()
            for _ in range(n_probe):
                step_fn(it, batches[it % nb], batches[(it + 1) % nb])
                it += 1
            sync()
            dt = (time.perf_counter() - t_) / n_probe * 1e3
            eng.stat_allreduce, eng.on_stage_grads_ready, trainer._dp_flush = keep_stat, keep_ready, keep_flush
            if hasattr(trainer.model, "on_backbone_grads_ready"):
                trainer.model.on_backbone_grads_ready = keep_buf
            return dt
        with_comm = probe(False)
        without = probe(True)
        t = torch.tensor([with_comm, without], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm = {"ms_per_step_with_comm": round(float(t[0]), 3), "ms_per_step_comm_stubbed": round(float(t[1]), 3),
                "comm_exposed_ms": round(float(t[0] - t[1]), 3), "steps": n_probe,
                "stage_group": int(os.environ.get("LIDK_DP_STAGE_GROUP", "3")),
                "note": "stubbed = gradient exchange skipped and SyncBatchNorm all-reduce replaced by a no-op (timing only)"}
    phases = phase_times(trainer, batches, step_fn)      # every rank runs these extra steps (they contain collectives)
    roof = roofline(trainer, batches, step_fn)
    it += 2
    if world > 1:
        dist.barrier()
    cavg = fit = cpu = None
    if args.cavg_steps > 0:
        cavg = cavg_phase(args, module, trainer, batches, step_fn, it)
    if args.fit_epochs > 0:
        log("Trainer.fit measurement")
        fit = fit_throughput(args, cfg, module, sets, trainer.dataloader_params, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_model, usable = host_info()
        log(f"cpu baseline on {usable} cores ({cpu_model})")
        if args.model in ("w2v2", "xlsr") or args.ragged:
            cpu = None                                   # the bounded CPU sample belongs to the two fixed-length headline workloads
        elif args.model == "wavlm":
            cpu = cpu_baseline_wavlm(module, ds, max(args.cpu_steps // 2, 3), 1, min(args.cpu_batch, 8), usable, cpu_model)
        else:
            cpu = cpu_baseline(module, ds, args.cpu_steps, 2, args.cpu_batch, usable, cpu_model)
    if rank == 0:
        audio_s = world * audio_s_rank                    # ragged: true audio of this rank's batches x ranks (same length mix)
        med = sorted(chunk_ms)[len(chunk_ms) // 2]
        if args.model in ("wavlm", "w2v2", "xlsr"):
            regime = {"heads": "backbone forward only (gradient stops at the features)",
                      "frozen": "encoder frozen as in the reference's first epochs: data gradients through the transformer to "
                                "layer_norm + mask_emb",
                      "finetune": "transformer encoder un-frozen: full encoder backward + Adam on it"}[args.wavlm_regime]
            dur = "1-10 s (1 s bins, bucketed padding; true audio counted)" if args.ragged else "3 s"
            if args.model == "wavlm":
                metric = f"audio-seconds/sec LID training, WavLM-base backbone ({args.wavlm_regime}) + Conformer heads"
                workload = (f"WavLMMutiLangModel: WavLM-base width backbone ({args.blocks} transformer layers, conv extractor on raw "
                            f"{dur}@16 kHz waveforms, span masking + dropouts on; {regime}) + {N_LANGS} Conformer CTC heads d768 "
                            f"(forward + backward), Adam, batch={args.batch}/GPU")
            elif args.model == "xlsr":
                metric = f"audio-seconds/sec joint CTC+LID training, XLS-R 300M (wav2vec2 Large) backbone ({args.wavlm_regime}) + Conformer heads"
                workload = (f"LidModule on Wav2vecMutiLangModel: XLS-R 300M architecture ({args.blocks} pre-LN layers d1024 / 16 heads / ffn "
                            f"4096, layer-norm conv extractor with bias, waveform layer-norm, key padding mask, last hidden state, span "
                            f"masking on; {regime}), {dur}@16 kHz waveforms, {N_LANGS} Conformer CTC heads d1024 (forward + backward), Adam, "
                            f"batch={args.batch}/GPU")
            else:
                metric = f"audio-seconds/sec joint CTC+LID training, wav2vec2-base backbone ({args.wavlm_regime}) + Conformer heads"
                workload = (f"LidModule on Wav2vecMutiLangModel: wav2vec2-base backbone ({args.blocks} transformer layers, key padding "
                            f"mask, hidden-state mix, span masking + dropouts on; {regime}), {dur}@16 kHz waveforms, {N_LANGS} Conformer "
                            f"CTC heads d768 (forward + backward), Adam, batch={args.batch}/GPU")
        else:
            metric = "audio-seconds/sec LID training, Conformer d256"
            dur = "1-10 s (1 s bins, bucketed padding; true audio counted)" if args.ragged else "3 s"
            workload = (f"ConformerLangModel {args.blocks}-layer d256, 14-lang CTC heads, log-mel 80-bin, {dur}@16 kHz, "
                        f"batch={args.batch}/GPU, Novograd+clip, features on GPU")
        line = {"metric": metric, "value": round(audio_s / elapsed, 1),
                "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": workload,
                           "global_batch": world * args.batch, "utterance_seconds": "1-10" if args.ragged else SECONDS,
                           "parallelism": f"dp{world}", "stochastic_depth": bool(args.stochastic_depth),
                           "grad_payload": trainer.grad_compress if world > 1 else None},
                "chunks_ms_per_step": [round(c, 3) for c in chunk_ms], "median_ms_per_step": round(med, 3),
                "value_at_median": round(audio_s / args.steps / (med * 1e-3), 1),
                "loss_first_last": [round(float(loss_vals[0]), 4), round(float(loss_vals[-1]), 4)], "loss_finite": finite,
                "host_issue_ms_per_step": round(host_issue * 1e3, 3), "comm": comm, "phases_ms": phases, "roofline": roof,
                # whole step against the MFMA roof: SURVEY 8d's algorithmic FLOPs (21.67 GFLOP per 3 s utterance: one head,
                # every block) over the timed wall clock - only meaningful for the headline Conformer workload
                "whole_step_tflops": (round(world * args.batch * FLOP_PER_UTT_TRAIN * args.steps / elapsed / 1e12, 1)
                                      if args.model == "conformer" and args.blocks == 12 and not args.stochastic_depth else None),
                "whole_step_mfma_frac": (round(world * args.batch * FLOP_PER_UTT_TRAIN * args.steps / elapsed / 1e12
                                               / (MFMA_BF16_PEAK_TFLOPS * world), 4)
                                         if args.model == "conformer" and args.blocks == 12 and not args.stochastic_depth else None),
                "val_cavg": cavg["val_cavg"] if cavg else None, "cavg": cavg, "fit": fit, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not finite:
        raise SystemExit("bench.py: a timed training step produced a non-finite loss")


if __name__ == "__main__":
    main()
