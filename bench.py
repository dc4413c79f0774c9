#!/usr/bin/env python3
"""Headline benchmark: Conformer-LID training throughput in audio-seconds/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the whole hot path over one batch of synthetic input already resident in HBM:
raw 3 s @ 16 kHz waveforms -> normalize + dither/pre-emphasis -> log-mel + SpecAugment (HIP) -> 12-layer d256 Conformer +
one 14-language CTC head, forward and backward (HIP, bf16 MFMA GEMMs) -> [N>1: gradient all-reduce over RCCL, SyncBN
statistics] -> global-norm clip(20) + Novograd + bf16 weight refresh (HIP) -> TriStage LR step.  Batch 64 per GPU (weak
scaling), BASELINE.json configs[1]; nothing is skipped or cached between steps.  Stochastic depth is OFF in the headline
number (every layer runs every step, matching the 21.67 GFLOP/utterance accounting of SURVEY 8d); pass
--stochastic-depth to time the reference default (p=0.7, ~16 % fewer executed blocks on average).

Prints ONE JSON line (rank 0) with the contract's fields plus:
  roofline     : the dominant kernel (gemm_nt: every Linear / 1x1 conv forward and data gradient; two instantiations of
                 the same 64x64-tile MFMA kernel, tile-pipelined for the wide K = 256 shapes):
                 algorithmic bytes of every launch of one training step / their summed durations, each launch bracketed by
                 HIP events on its own stream (HBM roofline: these K <= 1024 GEMMs are below the machine balance); the MFMA
                 rate of the same launches and the weight-gradient kernel are reported in the same object
  cpu_baseline : the CPU oracle (oracle/, torch fp32) running the same step on this host's cores, bounded sample
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


def build(args, rank, world, device):
    from ccml import seed_everything
    from ccml.trainer import Trainer
    from lid import hydra_lite
    import lid.main as launcher
    seed_everything(0)
    ov = [f"trainer.gpu_id={device.index}", "trainer.use_amp=true", f"trainer.ddp={'true' if world > 1 else 'false'}",
          f"trainer.world_size={world}", f"trainer.local_rank={rank}", "trainer.backend=nccl", "trainer.total_epoch=1000",
          f"data.sampler_common.train_batch_size={args.batch}", f"data.synthetic.items_per_lang={max(args.batch * world, 8)}",
          "data.synthetic.val_items_per_lang=2", "data.synthetic.test_items_per_lang=2", "module.interval=1000000",
          "trainer.log_interval=1000000", f"model.n_blocks={args.blocks}"]
    cfg = hydra_lite.load_config(os.path.join(ROOT, "speech-lid_amd", "lid", "conf"), "synthetic_cfg2", ov)
    module, sets, params = launcher.build(cfg, rank, world)
    module.model.use_stochastic_depth = bool(args.stochastic_depth)
    trainer = Trainer(callbacks=[], loggers=[], **dict(cfg["trainer"]))
    trainer.ccml_module = module
    trainer.train_dataset, trainer.val_dataset, trainer.test_dataset = sets["train"], sets["val"], sets["test"]
    trainer.dataloader_params = params
    module.point_trainer(trainer)
    trainer.trainer_prepare()
    trainer._zero_grad()
    return cfg, module, trainer, sets["train"]


def resident_batches(ds, rank, world, device, batch):
    """One batch per language, built once and moved to HBM (each rank gets different utterances of the same language)."""
    out = []
    per_lang = len(ds) // N_LANGS
    for k in range(N_LANGS):
        base = k * per_lang
        idx = [base + (rank * batch + j) % per_lang for j in range(batch)]
        b = list(ds.collate_fn([ds[i] for i in idx]))
        b[0] = b[0].to(device)
        for j in (1, 2, 3, 5):
            b[j] = b[j].to(device)
        out.append(b)
    return out


def cpu_baseline(module, ds, steps, batch, threads):
    """The oracle (torch-CPU fp32 restatement, pinned to the reference) running the same training step on host cores."""
    import random
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
    for step in range(steps + 1):
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
        if step > 0:
            times.append(time.time() - t0)
    t = sum(times) / len(times)
    return {"value": round(batch * SECONDS / t, 2), "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
            "sample": f"{steps} steps of batch {batch} (3 s utterances) after 1 warm-up step, {t:.2f} s/step, fp32 torch-CPU oracle"}


def _pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/r01/pmc_traffic.json, produced by
    tools/gpu_pmc_bench.sh on this same command): FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE counts 64 B per 128-B
    request on gfx950 and is doubled, as MI355X_MICROARCH.md (HBM) prescribes; WRITE_SIZE is exact for 16-byte stores."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    raw = json.load(open(path))
    hits = [v for name, v in raw.items() if kernel_prefix in name and "FETCH_SIZE_KB_per_launch_raw" in v
            and "WRITE_SIZE_KB_per_launch_raw" in v]
    n = sum(v.get("launches_fetch", 0) for v in hits)
    if not n:
        return None
    kb = sum((2.0 * v["FETCH_SIZE_KB_per_launch_raw"] + v["WRITE_SIZE_KB_per_launch_raw"]) * v["launches_fetch"] for v in hits)
    return round(kb / n * 1024.0)                                    # launch-weighted mean over the kernel's instantiations


def gemm_roofline(trainer, batches, step_fn):
    """Bracket every launch of the dominant kernel (gemm_nt: every Linear / 1x1 conv forward and its data gradient) of ONE
    training step with HIP events on the launch stream, and the weight-gradient GEMM (gemm_tn) likewise.  With K = 256..1024
    these GEMMs sit below the machine balance (about 150 FLOP per algorithmic byte against 2500 TF/s / 8 TB/s = 312), so the
    roofline that bounds them is HBM: achieved = algorithmic bytes (operands read once, outputs written once; DESIGN.md
    section 5) / measured time.  The MFMA rate of the same launches is reported beside it.  The step runs through the eager
    launch path (events cannot be recorded inside a hipGraph capture); kernels and shapes are those of the timed steps.  The
    cost of an empty event bracket is measured and subtracted."""
    eng = trainer.engine
    k = eng.k
    orig_nt, orig_tn = k.gemm_nt, k.gemm_tn
    rec = {"nt": [], "tn": []}

    def esz(t):
        return t.element_size()

    def bracket(kind, fn, flops, nbytes, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        rec[kind].append((e0, e1, flops, nbytes))
        return r

    def timed_nt(A, B, out, *a, M=None, N=None, K=None, **kw):
        m, kk, n = (A.shape[0] if M is None else M), (A.shape[1] if K is None else K), (B.shape[0] if N is None else N)
        nbytes = m * kk * esz(A) + n * kk * esz(B) + m * n * esz(out)
        for key in ("out2", "aux", "res"):
            if kw.get(key) is not None:
                nbytes += m * n * esz(kw[key])
        if kw.get("bias") is not None:
            nbytes += n * 4
        return bracket("nt", orig_nt, 2.0 * m * n * kk, nbytes, A, B, out, *a, M=M, N=N, K=K, **kw)

    def timed_tn(X, Y, C, *a, M=None, N1=None, N2=None, **kw):
        m, n1, n2 = (X.shape[0] if M is None else M), (X.shape[1] if N1 is None else N1), (Y.shape[1] if N2 is None else N2)
        nbytes = m * n1 * esz(X) + m * n2 * esz(Y) + n1 * n2 * 4
        return bracket("tn", orig_tn, 2.0 * m * n1 * n2, nbytes, X, Y, C, *a, M=M, N1=N1, N2=N2, **kw)

    graphs_on = eng.graphs.enabled
    eng.graphs.enabled = False
    k.gemm_nt, k.gemm_tn = timed_nt, timed_tn
    try:
        step_fn(0, batches[0])
        torch.cuda.synchronize()
    finally:
        k.gemm_nt, k.gemm_tn = orig_nt, orig_tn
        eng.graphs.enabled = graphs_on
    empty = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        empty.append((e0, e1))
    torch.cuda.synchronize()
    ovh = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]          # median, ms

    def tot(kind):
        ms = sum(max(a.elapsed_time(b) - ovh, 0.0) for a, b, _, _ in rec[kind])
        return ms, sum(f for _, _, f, _ in rec[kind]), sum(n for _, _, _, n in rec[kind]), len(rec[kind])

    ms, fl, by, n = tot("nt")
    ms2, fl2, by2, n2 = tot("tn")
    gbs = by / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": _pmc_traffic("gemm_nt_bf16_"), "kernel": "gemm_nt (gemm_nt_bf16_pipe_kernel + gemm_nt_bf16_direct_kernel, 64x64 tiles)",
            "launches_per_step": n, "avg_launch_us": round(ms * 1e3 / n, 2), "kernel_ms_per_step": round(ms, 3),
            "algorithmic_bytes_per_launch": round(by / n), "event_bracket_overhead_us": round(ovh * 1e3, 2),
            "mfma": {"achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
            "wgrad_kernel": {"kernel": "gemm_tn_bf16_kernel<64,64>", "launches_per_step": n2,
                             "avg_launch_us": round(ms2 * 1e3 / n2, 2), "GB/s": round(by2 / (ms2 * 1e-3) / 1e9, 1),
                             "TFLOP/s": round(fl2 / (ms2 * 1e-3) / 1e12, 1), "traffic": _pmc_traffic("gemm_tn_bf16_kernel")}}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--stochastic-depth", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-batch", type=int, default=16)
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the lidk path has no CPU fallback")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg, module, trainer, ds = build(args, rank, world, device)
    batches = resident_batches(ds, rank, world, device, args.batch)
    n_total = args.warmup + args.steps + 8

    def step_fn(i, batch, upcoming=None):
        trainer.train_step(i, batch, n_total, upcoming)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed set-up (the analogue of JIT compilation): each language head's block sequence is captured into a hipGraph on
    # its second use, so visit every language three times before the contract's W warm-up steps.  These are ordinary training
    # steps; nothing they compute is reused by the timed steps.
    if trainer.engine.graphs.enabled and not args.stochastic_depth:
        for i in range(3 * N_LANGS):
            step_fn(i, batches[i % N_LANGS])
    # From here on every step also starts the NEXT batch's feature kernels (normalize, dither, STFT/mel) on the feature
    # stream, as Trainer.fit does: each timed step still contains exactly one batch's worth of feature work.
    for i in range(args.warmup):
        step_fn(i, batches[i % N_LANGS], batches[(i + 1) % N_LANGS])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_fn(args.warmup + i, batches[(args.warmup + i) % N_LANGS], batches[(args.warmup + i + 1) % N_LANGS])
    host_issue = time.perf_counter() - t0          # host time to enqueue the K steps (no device sync inside)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    phases = phase_times(trainer, batches, step_fn)      # every rank runs these extra steps (they contain collectives)
    roof = gemm_roofline(trainer, batches, step_fn)
    if world > 1:
        dist.barrier()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 16)
        cpu = cpu_baseline(module, ds, args.cpu_steps, args.cpu_batch, threads)
    if rank == 0:
        audio_s = world * args.batch * SECONDS * args.steps
        line = {"metric": "audio-seconds/sec LID training, Conformer d256", "value": round(audio_s / elapsed, 1),
                "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"ConformerLangModel {args.blocks}-layer d256, 14-lang CTC heads, log-mel 80-bin, "
                                       f"3 s@16 kHz, batch={args.batch}/GPU, Novograd+clip, features on GPU",
                           "global_batch": world * args.batch, "utterance_seconds": SECONDS,
                           "parallelism": f"dp{world}", "stochastic_depth": bool(args.stochastic_depth)},
                "host_issue_ms_per_step": round(host_issue / args.steps * 1e3, 3), "phases_ms": phases, "roofline": roof,
                "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
