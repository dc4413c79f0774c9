"""Host-side orchestration of the HIP kernels for the Conformer-LID model (forward + hand-written backward).

One ``Engine`` owns, in HBM:
  * ``flat``  f32 master parameters, ``grad`` f32 gradients (parallel arenas, layout.py), BatchNorm buffers;
  * ``wT``    the GEMM operands in the activation dtype T (bf16 or f32): every weight W [N,K] and its transpose;
  * per-(B,T) workspaces: saved activations of every block (T) and the f32 residual stream.
torch provides memory and the current stream only; all arithmetic is lidk kernels.  There is no autograd inside:
``backward`` replays the block chain in reverse with explicit dgrad / wgrad GEMMs.

Reference semantics followed: lid/conformer.py:252-259 (block), :445-466 (encoder, stochastic depth),
lid/ConformerLangModel.py:272-294,352-356 (heads).
"""
import contextlib
import math
from typing import Callable, Dict, List, Optional

import torch

from . import _lib as L
from . import ops
from ._lib import LidkError
from .layout import ALIGN, ConformerCfg, Spec, init_values, model_specs


def _os_env(k, d):
    import os
    return os.environ.get(k, d)


def _ceil(a, b):
    return (a + b - 1) // b * b


class _GraphCache:
    """hipGraph capture of fixed launch sequences (one ConformerBlock forward or backward).

    A block's kernels always touch the same preallocated buffers and carry no per-step scalars, so the ~18 (forward) /
    ~35 (backward, including the previous block's weight gradients on the second stream) launches can be recorded once and
    replayed with a single host call: the Python/ctypes launch path costs ~13 us per kernel, a graph replay ~15 us per
    block.  First use of a key runs eagerly (allocations, lazy views), second use captures, later uses replay (with many
    distinct workspace shapes alive - a ragged corpus - a key must be seen three times before its capture is paid).  Disabled
    with LIDK_GRAPHS=0 and on the CPU test backend; under data parallelism a block's sequence is cut at the SyncBatchNorm
    all-reduce into two graphs (Engine._run_split).  A capture that fails switches the engine to eager launches."""

    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.state = {}
        self.eager_uses = 1        # eager runs of a key before it is captured (the engine raises it for ragged corpora)

    def run(self, key, fn):
        if not self.enabled:
            return fn()
        st = self.state.get(key)
        if st is None:
            self.state[key] = st = [None, 0]
        if st[0] is None and st[1] < self.eager_uses:
            st[1] += 1
            return fn()
        if st[0] is None:
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: other threads (e.g. the RCCL process group's watchdog polling its events) must
            # not invalidate a capture in progress
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn()
            except RuntimeError as err:                 # capture refused (nothing was executed): run this and all later
                import logging                          # sequences through the eager launch path instead
                logging.warning("lidk: hipGraph capture failed (%s); continuing without graphs", str(err).splitlines()[0])
                self.enabled = False
                self.state.clear()
                torch.cuda.synchronize()
                return fn()
            st[0] = g
        st[0].replay()

    def clear(self):
        self.state.clear()

    def drop(self, work_id: int):
        """Forget the graphs recorded against one workspace (every key carries id(workspace) in position 1)."""
        for k in [k for k in self.state if len(k) > 1 and k[1] == work_id]:
            del self.state[k]


class _BlockParams:
    """Views of one ConformerBlock's parameters: f32 master / gradient views and T-typed GEMM operands."""

    def __init__(self, eng: "Engine", p: str, heads: int, dh: int):
        self.p, self.heads, self.dh = p, heads, dh
        v, g, w = eng.pview, eng.gview, eng.wview
        self.names = [s.name for s in eng.specs if s.name.startswith(p + ".")]
        for tag, n in (("ff1", ".ff1.fn"), ("ff2", ".ff2.fn")):
            setattr(self, tag, dict(
                ln_w=v(p + n + ".norm.weight"), ln_b=v(p + n + ".norm.bias"),
                dln_w=g(p + n + ".norm.weight"), dln_b=g(p + n + ".norm.bias"),
                w1=w(p + n + ".fn.net.0.weight"), b1=v(p + n + ".fn.net.0.bias"),
                dw1=g(p + n + ".fn.net.0.weight"), db1=g(p + n + ".fn.net.0.bias"),
                w2=w(p + n + ".fn.net.3.weight"), b2=v(p + n + ".fn.net.3.bias"),
                dw2=g(p + n + ".fn.net.3.weight"), db2=g(p + n + ".fn.net.3.bias")))
        a = p + ".attn"
        self.attn = dict(ln_w=v(a + ".norm.weight"), ln_b=v(a + ".norm.bias"),
                         dln_w=g(a + ".norm.weight"), dln_b=g(a + ".norm.bias"),
                         wqkv=eng.wview_qkv(a + ".fn"), dwqkv=eng.gview_qkv(a + ".fn"),
                         wo=w(a + ".fn.to_out.weight"), bo=v(a + ".fn.to_out.bias"),
                         dwo=g(a + ".fn.to_out.weight"), dbo=g(a + ".fn.to_out.bias"),
                         emb=v(a + ".fn.rel_pos_emb.weight"), demb=g(a + ".fn.rel_pos_emb.weight"),
                         embT=eng.wview(a + ".fn.rel_pos_emb.weight")[0])
        c = p + ".conv.net"
        self.conv = dict(ln_w=v(c + ".0.weight"), ln_b=v(c + ".0.bias"), dln_w=g(c + ".0.weight"), dln_b=g(c + ".0.bias"),
                         w1=w(c + ".2.weight"), b1=v(c + ".2.bias"), dw1=g(c + ".2.weight"), db1=g(c + ".2.bias"),
                         dw=v(c + ".4.conv.weight"), dwb=v(c + ".4.conv.bias"),
                         ddw=g(c + ".4.conv.weight"), ddwb=g(c + ".4.conv.bias"),
                         bn_w=v(c + ".5.weight"), bn_b=v(c + ".5.bias"), dbn_w=g(c + ".5.weight"), dbn_b=g(c + ".5.bias"),
                         rm=eng.buffers[c + ".5.running_mean"], rv=eng.buffers[c + ".5.running_var"],
                         nbt=eng.buffers[c + ".5.num_batches_tracked"],
                         w2=w(c + ".7.weight"), b2=v(c + ".7.bias"), dw2=g(c + ".7.weight"), db2=g(c + ".7.bias"))
        self.post = dict(w=v(p + ".post_norm.weight"), b=v(p + ".post_norm.bias"),
                         dw=g(p + ".post_norm.weight"), db=g(p + ".post_norm.bias"))


class _Pool:
    """HBM behind the workspaces of one capacity class (batch, frames rounded up to a bucket).  Ragged corpora produce
    hundreds of distinct (B, F) shapes; each gets its own ``_Work`` (a set of VIEWS with the exact M and T the kernels need)
    carved out of the class's pool, so a new shape costs a few hundred view objects instead of ~3 GB of fresh allocations.
    ``measure`` mode hands out meta tensors and records the bytes a workspace of the capacity shape needs."""

    def __init__(self, device):
        self.device, self.store, self.cursor, self.need, self.measure = device, {}, {}, {}, True

    def take(self, shape, dtype, zero=False):
        n = int(math.prod(shape)) if shape else 1
        n_al = _ceil(max(n, 1), 64)
        cur = self.cursor.get(dtype, 0)
        self.cursor[dtype] = cur + n_al
        if self.measure:
            self.need[dtype] = self.cursor[dtype]
            return torch.empty(*shape, device="meta", dtype=dtype)
        t = self.store[dtype][cur:cur + n].view(*shape)
        if zero:
            t.zero_()
        return t

    def commit(self):
        self.store = {dt: torch.empty(n, device=self.device, dtype=dt) for dt, n in self.need.items()}
        self.measure = False

    def rewind(self):
        self.cursor = {}


class _BlockBuf:
    """Saved activations of one block for one (B, T)."""

    def __init__(self, eng: "Engine", B: int, T: int, heads: int, dh: int, ff: int, ci: int, pool: _Pool):
        M, d, dt = B * T, eng.cfg.d, eng.act_dtype
        e = lambda *s: pool.take(s, dt)
        f = lambda *s: pool.take(s, torch.float32)
        inner = heads * dh
        self.mean = [f(M) for _ in range(5)]
        self.rstd = [f(M) for _ in range(5)]
        self.h1, self.a1, self.u1, self.x1 = e(M, d), e(M, ff), e(M, ff), f(M, d)
        self.h2, self.qkv, self.o, self.x2 = e(M, d), e(M, 3 * inner), e(M, inner), f(M, d)
        # attention probabilities [B, heads, T, ldp]: only for shapes whose backward cannot recompute them (the MFMA kernels with
        # T <= 256 keep a log-sum-exp per row instead - ~260 MB per block at B = 64, T ~ 500 would otherwise sit unused)
        self.ldp = eng.k.attn_ldp(T, dh, dt)
        self.probs = None if eng._attn_recompute(T, dh) else e(B, heads, T, self.ldp)
        self.h3, self.y, self.g, self.c, self.s, self.x3 = e(M, d), e(M, 2 * ci), e(M, ci), e(M, ci), e(M, ci), f(M, d)
        self.bn_mean, self.bn_rstd = f(ci), f(ci)
        self.h4, self.a4, self.u4, self.x4 = e(M, d), e(M, ff), e(M, ff), f(M, d)
        self.out = f(M, d)


class _Work:
    """Everything sized by (B, F): front-end buffers, per-block buffers, backward scratch."""

    def __init__(self, eng: "Engine", B: int, F_: int, pool: _Pool):
        cfg = eng.cfg
        self.B, self.F, self.pool = B, F_, pool
        self.T = T = eng.frames_to_steps(F_)
        self.M = M = B * T
        self.Mp = Mp = _ceil(M, 8)
        d, dev, dt = cfg.d, eng.device, eng.act_dtype
        pool.rewind()
        e = lambda *s: pool.take(s, dt)
        z = lambda *s: pool.take(s, dt, zero=True)
        f = lambda *s: pool.take(s, torch.float32)
        f64 = lambda *s: pool.take(s, torch.float64)
        u8 = lambda *s: pool.take(s, torch.uint8)
        ff, ci = d * cfg.ff_mult, d * cfg.conv_expansion_factor
        self.col = e(M, 3 * cfg.n_mels)
        ff_e, ci_e = ff, ci
        self.r = e(M, cfg.n_mels)
        self.x0 = f(M, d)
        self.x0d = f(M, d)
        self.pos_keep = u8(M * d)
        self.enc = [_BlockBuf(eng, B, T, cfg.heads, cfg.dim_head, ff_e, ci_e, pool) for _ in range(cfg.n_blocks)]
        ff, ci = max(ff_e, 4 * d), max(ci_e, 2 * d)          # scratch must also fit the head block (ff 4d, conv 2d, k 31)
        self.head = _BlockBuf(eng, B, T, cfg.last_heads, cfg.last_dim_head, d * 4, d * 2, pool)
        self.head_h = e(M, d)
        self.head_keep = u8(M * d)
        self.v1p = _ceil(max(cfg.lang2vocab.values()) + 1, 8)
        self.dlT = z(M, self.v1p)
        # backward scratch (shared by all blocks)
        cmax = max(ff, 2 * ci, 3 * cfg.heads * cfg.dim_head, 3 * cfg.last_heads * cfg.last_dim_head, 4 * d, 3 * cfg.n_mels,
                   self.v1p)
        self.cmax = cmax
        # Each weight-gradient site of a block backward reads its own dY buffer, and consecutive blocks alternate between two
        # such sets: the weight-gradient GEMMs of a block can then run beside the NEXT block's dgrad chain (deferred mode,
        # Engine.backward) without either side overwriting what the other still reads.
        class _Set:
            pass
        self.sets = []
        # LIDK_SCRATCH_SETS=3: a third set lets the whole-chain backward leave the weight-gradient stream alone until the end of
        # a block (with two sets the fused LayerNorm pair at a block's tail writes into the set the previous block's weight
        # gradients are still reading, so the data-gradient chain joins the second stream before it: a second cross-queue hop per
        # block in the captured graph).  Round 3: 7.39 / 7.30 vs 7.39 / 7.36 ms per step - within noise; round 4, with the
        # data-gradient chain as the critical path: 6.852 / 6.854 vs 6.89 / 6.90 (same box, two rounds each): the default is 3.
        for _ in range(max(2, int(_os_env("LIDK_SCRATCH_SETS", "3")))):
            S = _Set()
            S.da = [e(M, ff), e(M, ff)]                       # ff2 / ff1 hidden gradients
            S.dy1 = e(M, 2 * ci)                              # conv pointwise-1 output gradient
            S.dqkv = e(M, 3 * max(cfg.heads * cfg.dim_head, cfg.last_heads * cfg.last_dim_head))
            S.dyTs = [e(M, d) for _ in range(4)]              # T-typed dx at: block output (x0.5), x3, x2, x1
            S.dc = e(M, ci)                                   # depthwise-conv output gradient (for its weight gradient)
            S.ds = e(M, ci)                                   # gradient at the BatchNorm+Swish output
            # BN backward sums (all ranks / this rank); element [2*ci] carries the row count, which the same all-reduce
            # turns into the global count (ranks may hold different (B, T) shapes)
            S.sums = f64(2 * ci + 1)
            S.sums_local = f64(2 * ci + 1)
            S.dsc = f(B, max(cfg.heads, cfg.last_heads), T, (T + 31) // 32 * 32)     # attention dS rows (kept for the deferred dE)
            S.lnp = [f(L.LN_BWD_BLOCKS * 2 * d) for _ in range(5)]   # LayerNorm dgamma/dbeta partial rows: post, ff2, conv, attn, ff1
            self.sets.append(S)
        self.dmid = e(M, max(ci, cfg.heads * cfg.dim_head, cfg.last_heads * cfg.last_dim_head, d))   # ds / do
        self.dh = e(M, d)
        self.dyT = self.sets[0].dyTs[0]                       # the head block (first in backward) uses set 0
        self.dxa, self.dxb = f(M, d), f(M, d)
        hmax = max(cfg.heads, cfg.last_heads)
        self.partial = f(max(L.LN_PARTIAL_BLOCKS * 2 * cmax, L.LN_BWD_BLOCKS * 2 * d, L.BN_PARTIAL_BLOCKS * 2 * ci))
        self.stat_parts = eng.k.dwconv_stat_parts(B, T, ci, eng.act_dtype)
        self.stat_partial = f(self.stat_parts * 2 * ci)
        self.dw_partial = f(B * ci * (max(cfg.conv_kernel_size, 31) + 1))
        self.sums = f64(2 * ci + 1)                                                 # (sum x, sum x^2) [2*ci] + row count
        self.dconv3 = f(cfg.n_mels, 3 * cfg.n_mels)
        self.dfeat = f(M, d)
        self.logits: Dict[str, torch.Tensor] = {}


_WGRAD_WGS = int(__import__("os").environ.get("LIDK_WGRAD_WGS", "512"))      # workgroups a weight-gradient launch aims for (tuning knob)
# Grouped weight-gradient launch: output tile and row chunks per weight gradient.  Measured end to end at cfg2 (same box, two rounds
# each, ms per step): 64-tiles x 4 chunks 7.73 (x 8: 7.86); 128-tiles x 1 / 2 / 3 / 4 / 6 / 8 chunks 7.93 / 7.45 / 7.61 / 7.57 /
# 7.70 / 7.70 - twice the MFMA work per operand byte fetched from L2 (the 64-tiles pull ~1 GB through L2 per launch) and ~190
# long items instead of ~1 700 short ones beside the data-gradient chain.  With the LDS-DMA ring (csrc/gemm.hip) the split no longer matters:
# 128-tiles x 2 / 3 / 4 chunks 7.17 / 7.16 / 7.20 (two rounds, same box).
# Whole-chain backward capture: record the fork point before a block's data-gradient chain but capture the forked weight-gradient work
# AFTER it (same dependencies).  The chain's first kernel is then the first child of the previous block's last kernel in the
# hipGraph and the chain stays on one hardware queue (before: the last LayerNorm backward of every block ran on the side queue,
# a cross-queue hand-over of ~10 us on either side of it).  7.20 -> 7.14 ms per step at cfg2 (two rounds, same box).
_FORK_LATE = __import__("os").environ.get("LIDK_FORK_LATE", "1") == "1"
_WGRAD_TILE = int(__import__("os").environ.get("LIDK_WGRAD_TILE", "128"))
_WGRAD_SPLIT = int(__import__("os").environ.get("LIDK_WGRAD_SPLIT", "2" if _WGRAD_TILE == 128 else "4"))


class Engine:
    def __init__(self, cfg: ConformerCfg, act_dtype=torch.bfloat16, backend=None):
        self.cfg = cfg
        self.act_dtype = act_dtype
        self._after_first = None
        # kernel backend: the real lidk.ops (HIP; refuses CPU tensors).  tests/ may inject a torch-CPU fake to check
        # the orchestration on a GPU-less machine; nothing in the product constructs an Engine with another backend.
        self.k = backend if backend is not None else ops
        self._hip = bool(getattr(self.k, "IS_HIP_BACKEND", False))
        self.device = torch.device("cpu")
        self.specs, self.buffer_specs, self.stages, self.n_flat = model_specs(cfg)
        self.by_name: Dict[str, Spec] = {s.name: s for s in self.specs}
        self.flat = torch.zeros(self.n_flat)
        self.grad: Optional[torch.Tensor] = None
        self.buffers: Dict[str, torch.Tensor] = {}
        for name, shape, dt in self.buffer_specs:
            self.buffers[name] = torch.ones(shape) if name.endswith("running_var") else torch.zeros(shape, dtype=dt)
        self._built = False
        self.reset_parameters()
        self._work: Dict[tuple, _Work] = {}
        self._pools: Dict[tuple, _Pool] = {}
        import os as _os
        self.graphs = _GraphCache(self._hip and _os.environ.get("LIDK_GRAPHS", "1") != "0")
        # data-parallel hooks (set by the Trainer): all-reduce of f64 BatchNorm sums, and "gradients of stage ready"
        self.stat_allreduce: Optional[Callable[[torch.Tensor], None]] = None
        self.on_stage_grads_ready: Optional[Callable[[str], None]] = None
        self.world_size = 1
        self.seed = 0                      # the run's seed (Trainer._seed_native_generators): equal on every rank
        self.rank_salt = 0                 # data parallelism: this rank's salt of the DROPOUT masks (stochastic depth uses seed only)
        self.step_count = 0
        self._split_ok: Dict[tuple, bool] = {}
        self._ln_gemm_ok: Dict[tuple, bool] = {}
        # Second HIP stream for the weight-gradient GEMMs (LIDK_SIDE_STREAM=0 turns it off): a block's wgrads are deferred and
        # run beside the next block's dgrad chain, one fork/join per captured block graph: 9.75 vs 10.01 ms/step.  (Forking
        # at every wgrad site cost more in graph edges than the overlap won: 13.0 vs 11.85 ms/step at the time.)
        self.side = None

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self):
        vals = init_values(self.cfg)
        for s in self.specs:
            self.flat[s.offset:s.offset + s.numel] = vals[s.name].reshape(-1).to(self.flat.device)
        for name, t in self.buffers.items():
            if name.endswith("running_var"):
                t.fill_(1)
            else:
                t.zero_()
        if self._built:
            self.refresh_weights()

    def to(self, device):
        device = torch.device(device)
        if device.type == "cpu" and not self._built and self._hip:
            return self                      # still on the host before the first move to a GPU: nothing to build
        self.device = device
        self.flat = self.flat.to(device)
        self.grad = torch.zeros_like(self.flat)
        self.buffers = {k: v.to(device) for k, v in self.buffers.items()}
        self._build_operands()
        if self._hip and device.type == "cuda" and _os_env("LIDK_SIDE_STREAM", "1") != "0":
            self.side = torch.cuda.Stream(device=device)
        return self

    def load_state(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Copy a reference-named state dict (parameters + BatchNorm buffers) into the arenas."""
        missing = [n for n in list(self.by_name) + list(self.buffers) if n not in sd]
        unexpected = [n for n in sd if n not in self.by_name and n not in self.buffers]
        if strict and (missing or unexpected):
            raise KeyError(f"state dict mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for n, t in sd.items():
                if n in self.by_name:
                    s = self.by_name[n]
                    if tuple(t.shape) != tuple(s.shape):
                        raise ValueError(f"{n}: shape {tuple(t.shape)} != {tuple(s.shape)}")
                    self.flat[s.offset:s.offset + s.numel].copy_(t.reshape(-1).to(self.flat.dtype))
                elif n in self.buffers:
                    self.buffers[n].copy_(t.to(self.buffers[n].dtype))
        if self._built:
            self.refresh_weights()

    def state(self) -> Dict[str, torch.Tensor]:
        out = {s.name: self.pview(s.name) for s in self.specs}
        out.update(self.buffers)
        return out

    def zero_grad(self):
        self.grad.zero_()

    def pview(self, name):
        s = self.by_name[name]
        return self.flat[s.offset:s.offset + s.numel].view(s.shape)

    def gview(self, name):
        s = self.by_name[name]
        return self.grad[s.offset:s.offset + s.numel].view(s.shape)

    def wview(self, name):
        """(W [N,K], W^T [K,ldt]) views in the T arena."""
        return self._wviews[name]

    def wview_qkv(self, prefix):
        return self._wviews[prefix + ".to_q.weight+to_kv"]

    def gview_qkv(self, prefix):
        q, kv = self.by_name[prefix + ".to_q.weight"], self.by_name[prefix + ".to_kv.weight"]
        assert kv.offset == q.offset + q.numel, "to_q/to_kv must be adjacent in the arena"
        n = q.shape[0] + kv.shape[0]
        return self.grad[q.offset:q.offset + n * q.shape[1]].view(n, q.shape[1])

    def _build_operands(self):
        """Lay out the T-typed copies: per GEMM weight W [N,K] and W^T [K, ceil8(N)] (zero padded)."""
        mats, views, off = [], {}, 0

        def add(key, src_off, n, k):
            nonlocal off
            ldt = _ceil(n, 8)
            w_off, off = off, off + _ceil(n * k, ALIGN)
            t_off, off = off, off + _ceil(k * ldt, ALIGN)
            mats.append([src_off, n, k, w_off, t_off, ldt])
            views[key] = (w_off, t_off, n, k, ldt)

        skip = set()
        for s in self.specs:
            if s.kind == "emb":                                   # T-typed copy of the relative embeddings (MFMA attention)
                w_off, off = off, off + _ceil(s.numel, ALIGN)
                mats.append([s.offset, s.shape[0], s.shape[1], w_off, -1, 0])
                views[s.name] = (w_off, -1, s.shape[0], s.shape[1], 0)
                continue
            if s.kind != "w" or s.name in skip:
                continue
            if s.name.endswith(".to_q.weight"):
                kv = self.by_name[s.name.replace("to_q", "to_kv")]
                assert kv.offset == s.offset + s.numel
                add(s.name[:-len(".to_q.weight")] + ".to_q.weight+to_kv", s.offset, s.shape[0] + kv.shape[0], s.shape[1])
                skip.add(kv.name)
                continue
            add(s.name, s.offset, s.shape[0], s.shape[1])
        c3 = self.by_name.get("model.featurizer.sub_sampling.sub_sampling.0.weight")
        if c3 is None:
            c3 = Spec("none", (8, 1, 1), "conv3")                   # features front: no subsampling conv (placeholder operand)
        self._c3_off, off = off, off + _ceil(c3.shape[0] * 3 * c3.shape[1], ALIGN)
        self.wT = torch.zeros(off, device=self.device, dtype=self.act_dtype)
        self.mats, self.mat_tiles = self.k.build_cast_table(mats, self.device)
        self._mat_rows, self._lang_tables = mats, {}
        self._wviews = {}
        for key, (w_off, t_off, n, k, ldt) in views.items():
            self._wviews[key] = (self.wT[w_off:w_off + n * k].view(n, k),
                                 self.wT[t_off:t_off + k * ldt].view(k, ldt) if t_off >= 0 else None)
        self.w_conv3 = self.wT[self._c3_off:self._c3_off + c3.shape[0] * 3 * c3.shape[1]].view(c3.shape[0], 3 * c3.shape[1])
        fz = "model.featurizer"
        self.enc_params = [_BlockParams(self, f"{fz}.encoders.{i}", self.cfg.heads, self.cfg.dim_head)
                           for i in range(self.cfg.n_blocks)]
        self.head_params = {l: _BlockParams(self, f"model.last_projects.{l}.block", self.cfg.last_heads, self.cfg.last_dim_head)
                            for l in self.cfg.lang2vocab}
        self._built = True
        self._work.clear()
        self._pools.clear()
        self.graphs.clear()
        self.refresh_weights()

    def refresh_weights(self, lang: Optional[str] = None):
        """f32 master -> T operands (after an optimizer step, load_state_dict or reset).  With ``lang`` only the tensors a
        training step for that language can have changed are converted (encoder, front end, that language's head): the
        other 13 heads are half of the arena."""
        if lang is None:
            self.k.cast_weights(self.flat, self.wT, self.mats, self.mat_tiles)
        else:
            if lang not in self._lang_tables:
                other = [self.stage_range(f"head.{l}") for l in self.cfg.lang2vocab if l != lang]
                rows = [r for r in self._mat_rows if not any(lo <= r[0] < hi for lo, hi in other)]
                self._lang_tables[lang] = self.k.build_cast_table(rows, self.device)
            mats, tiles = self._lang_tables[lang]
            self.k.cast_weights(self.flat, self.wT, mats, tiles)
        if self.cfg.front == "subsample":
            c3 = self.pview("model.featurizer.sub_sampling.sub_sampling.0.weight")       # [Co][Ci][3] -> [Co][k*Ci+ci]
            self.w_conv3.copy_(c3.permute(0, 2, 1).reshape(c3.shape[0], -1))             # 19 K elements: layout glue, not math

    def work(self, B, F_):
        """Workspace for a (batch, frames) shape.  Ragged training data produces hundreds of distinct shapes, so HBM is
        budgeted per CAPACITY CLASS (batch, frames rounded up to LIDK_FRAME_BUCKET = 64): a class owns one pool sized for its
        largest shape (HBM is 288 GB, a cfg2 pool is ~3 GB; LIDK_MAX_WORKSPACES = 12 classes stay resident, least recently
        used goes first) and every exact (B, F) gets a cheap set of views into it with the M and T the kernels need.  Evicting
        a class drops its view sets and only the hipGraphs recorded against them."""
        key = (B, F_)
        w = self._work.pop(key, None)
        if w is None:
            if self._hip and hasattr(self.k, "attn_max_frames"):
                T = self.frames_to_steps(F_)
                lim = min(self.k.attn_max_frames(self.cfg.dim_head, self.act_dtype),
                          self.k.attn_max_frames(self.cfg.last_dim_head, self.act_dtype))
                if T > lim:
                    raise LidkError(f"a batch of {F_} feature frames (T = {T} after subsampling, {F_ / 100:.1f} s of audio) exceeds the "
                                    f"attention kernels' limit of T = {lim} ({lim / 50:.0f} s): lower data.max_duration")
            bucket = max(int(_os_env("LIDK_FRAME_BUCKET", "64")), 1)
            ckey = (B, _ceil(F_, bucket))
            pool = self._pools.pop(ckey, None)
            if pool is None:
                limit = max(int(_os_env("LIDK_MAX_WORKSPACES", "12")), 1)
                while len(self._pools) >= limit:
                    old_key = next(iter(self._pools))
                    self._drop_pool(self._pools.pop(old_key))
                pool = _Pool(self.device)
                _Work(self, ckey[0], ckey[1], pool)            # measure the capacity shape (meta tensors, no HBM touched)
                pool.commit()
            self._pools[ckey] = pool
            while len(self._work) >= 256:                      # view sets are cheap, but not free
                old = self._work.pop(next(iter(self._work)))
                self.graphs.drop(id(old))
            w = _Work(self, B, F_, pool)
            self.graphs.eager_uses = 1 if len(self._work) < 8 else 3      # ragged corpus: most shapes never come back
        else:
            for ckey, pool in list(self._pools.items()):       # keep the class of a reused shape at the recent end
                if pool is w.pool:
                    self._pools[ckey] = self._pools.pop(ckey)
                    break
        self._work[key] = w                            # (re)insert at the most-recently-used end
        return w

    def _drop_pool(self, pool):
        for k in [k for k, w in self._work.items() if w.pool is pool]:
            self.graphs.drop(id(self._work.pop(k)))    # captured graphs point into the evicted pool

    def frames_to_steps(self, F_: int) -> int:
        """Sequence length the blocks see for an input of F_ frames: Conv1d(k3, s2, p1) subsampling, or the input itself when
        the engine is fed the features of a frozen backbone."""
        return (F_ + 2 - 3) // 2 + 1 if self.cfg.front == "subsample" else F_

    # ------------------------------------------------------------------ forward pieces
    def _front_fwd(self, w: _Work, mel, training, seed):
        cfg = self.cfg
        if cfg.front != "subsample":                 # backbone features (B, T, d): the residual stream starts from them; copied
            self.k.scale_cast(mel.view(w.M, cfg.d), w.x0, 1.0)       # into the workspace so captured graphs see one address
            return w.x0
        fz = "model.featurizer.sub_sampling"
        self.k.im2col_k3s2(mel, w.col, w.T)
        self.k.gemm_nt(w.col, self.w_conv3, w.r, bias=self.pview(fz + ".sub_sampling.0.bias"), act=L.ACT_RELU)
        self.k.gemm_nt(w.r, self.wview(fz + ".linear.weight")[0], w.x0, bias=self.pview(fz + ".linear.bias"),
                    alpha=math.sqrt(cfg.d))
        if training and cfg.pos_dropout > 0:
            self.k.dropout(w.x0, w.x0d, cfg.pos_dropout, seed=seed, keep_in=self._forced_masks.get("pos"),
                        keep_out=w.pos_keep)
            return w.x0d
        return w.x0

    def _ln2_ok(self) -> bool:
        """The fused LayerNorm pair (post_norm of block i + first PreNorm of block i + 1, forward and backward) is available."""
        return bool(self._hip and self.cfg.d <= 256 and hasattr(self.k, "layernorm2_fwd") and _os_env("LIDK_LN2", "1") == "1")

    def _ln_gemm(self, x, P, W, out, h, mean, rstd, ln_done: bool = False, **epi):
        """PreNorm + the projection that consumes it: two launches.  LIDK_LN_GEMM=1 opts into the row-panel kernel that does
        both in one (LayerNorm in the operand load); measured SLOWER on MI355X (LN + ff-up 41.7 vs 23.7 us, LN + QKV 20.6 vs
        17.3 us: two 64 KB workgroups per CU hide less latency than the per-tile kernels' 16+ waves), so it is off by default
        and kept for the record (measured and rejected)."""
        if ln_done:                      # h / mean / rstd were written by the previous block's fused LayerNorm pair
            self.k.gemm_nt(h, W, out, **epi)
            return
        M, N = x.shape[0], W.shape[0]
        key = (M, N)
        ok = self._ln_gemm_ok.get(key)
        if ok is None:
            ok = self._ln_gemm_ok[key] = bool(
                self._hip and _os_env("LIDK_LN_GEMM", "0") == "1" and hasattr(self.k, "ln_gemm_supported")
                and self.k.ln_gemm_supported(M, N, x.shape[1], self.act_dtype))
        if ok:
            self.k.ln_gemm_nt(x, P["ln_w"], P["ln_b"], W, out, h=h, mean=mean, rstd=rstd, **epi)
        else:
            self.k.layernorm_fwd(x, P["ln_w"], P["ln_b"], yT=h, mean=mean, rstd=rstd)
            self.k.gemm_nt(h, W, out, **epi)

    def _ffn_fused(self, M: int, ff: int) -> bool:
        """The whole FeedForward module as one launch forward and one backward (csrc/ffn.hip: d = 256, bf16; LIDK_FFN_FUSED=0 restores
        the LayerNorm / GEMM / GEMM sequences)."""
        key = ("ffn", M, ff)
        ok = self._split_ok.get(key)
        if ok is None:
            # the backward writes one partial (dgamma | dbeta) row per 64 input rows into a buffer of LN_BWD_BLOCKS rows
            ok = self._split_ok[key] = bool(self._hip and hasattr(self.k, "ffn_fwd") and _os_env("LIDK_FFN_FUSED", "1") == "1"
                                            and self.k.ffn_fwd_supported(M, self.cfg.d, ff, self.act_dtype)
                                            and self.k.ffn_bwd_partial_rows(M) <= L.LN_BWD_BLOCKS)
        return ok

    def _ff_fwd(self, x, P, h, a, u, xo, mean, rstd, ln_done: bool = False, next_ln=None) -> bool:
        """One FeedForward module.  next_ln: the LayerNorm(s) that consume its output (ops.ffn_fwd's ``next_ln``); the fused kernel
        applies them in its epilogue and the function returns True - otherwise the caller launches them."""
        if self._ffn_fused(x.shape[0], a.shape[1]):
            if _os_env("LIDK_FFN_NEXT_LN", "1") != "1":
                next_ln = None
            if ln_done:                  # h / mean / rstd were written by the previous block's fused LayerNorm pair
                self.k.ffn_fwd(x, P["w1"][0], P["b1"], P["w2"][0], P["b2"], xo, h_in=h, a=a, u=u, alpha=0.5, next_ln=next_ln)
            else:
                self.k.ffn_fwd(x, P["w1"][0], P["b1"], P["w2"][0], P["b2"], xo, gamma=P["ln_w"], beta=P["ln_b"], h=h, mean=mean,
                               rstd=rstd, a=a, u=u, alpha=0.5, next_ln=next_ln)
            return next_ln is not None
        self._ln_gemm(x, P, P["w1"][0], u, h, mean, rstd, ln_done=ln_done, bias=P["b1"], act=L.ACT_SWISH, out2=a)
        self.k.gemm_nt(u, P["w2"][0], xo, bias=P["b2"], alpha=0.5, res=x)
        return False

    def _block_fwd(self, x, bp: _BlockParams, bb: _BlockBuf, w: _Work, training: bool, part: str = "all",
                   ff1_ln_done: bool = False, tail=None) -> bool:
        """One ConformerBlock up to x4 (before post_norm).  ``part`` splits the launch sequence at the SyncBatchNorm
        collective: 'a' = up to the BatchNorm partial sums, 'b' = from the BatchNorm statistics on, 'all' = both."""
        B, T, M = w.B, w.T, w.M
        C = bp.conv
        ci, K = C["dw"].shape[0], C["dw"].shape[2]
        if part in ("all", "a"):
            A = bp.attn
            # the attention's PreNorm is applied in the first FeedForward's epilogue where the fused kernel runs
            ln_a = self._ff_fwd(x, bp.ff1, bb.h1, bb.a1, bb.u1, bb.x1, bb.mean[0], bb.rstd[0], ln_done=ff1_ln_done,
                                next_ln=dict(gA=A["ln_w"], bA=A["ln_b"], yAT=bb.h2, meanA=bb.mean[1], rstdA=bb.rstd[1]))
            self._ln_gemm(bb.x1, A, A["wqkv"][0], bb.qkv, bb.h2, bb.mean[1], bb.rstd[1], ln_done=ln_a)
            self.k.attn_fwd(bb.qkv, A["emb"], bb.o, None if self._attn_recompute(T, bp.dh) else bb.probs, B, T, bp.heads, bp.dh,
                            rel_emb_T=A["embT"])
            self.k.gemm_nt(bb.o, A["wo"][0], bb.x2, bias=A["bo"], res=bb.x1)
            self._ln_gemm(bb.x2, C, C["w1"][0], bb.y, bb.h3, bb.mean[2], bb.rstd[2], bias=C["b1"])
            dw2d = C["dw"].view(ci, K)
            pad_left = K // 2
            if training:                   # GLU fused into the depthwise conv's tile load; g is kept for the weight gradient
                self.k.glu_dwconv_fwd(bb.y, dw2d, C["dwb"], bb.g, bb.c, w.stat_partial, B, T, pad_left)
                if not self._fused_bn_stats():
                    self.k.reduce_partials_f64(w.stat_partial, w.stat_parts, 2 * ci, w.sums[:2 * ci + 1], tail=M)
            else:
                self.k.glu_dwconv_fwd(bb.y, dw2d, C["dwb"], None, bb.c, None, B, T, pad_left)
        if part in ("all", "b"):
            if training and self._fused_bn_stats():        # single process: partial rows -> statistics in one launch
                self.k.bn_train_stats_from_partials(w.stat_partial, w.stat_parts, M, bb.bn_mean, bb.bn_rstd, C["rm"], C["rv"],
                                                    C["nbt"])
            elif training:
                # count 0: read the (all-reduced) row count from w.sums[2*ci]
                self.k.bn_train_stats(w.sums[:2 * ci + 1], 0, bb.bn_mean, bb.bn_rstd, C["rm"], C["rv"], C["nbt"])
            else:
                self.k.bn_eval_stats(C["rm"], C["rv"], bb.bn_mean, bb.bn_rstd)
            self.k.bn_swish_fwd(bb.c, bb.bn_mean, bb.bn_rstd, C["bn_w"], C["bn_b"], bb.s)
            self.k.gemm_nt(bb.s, C["w2"][0], bb.x3, bias=C["b2"], res=bb.x2)
            return self._ff_fwd(bb.x3, bp.ff2, bb.h4, bb.a4, bb.u4, bb.x4, bb.mean[3], bb.rstd[3], next_ln=tail)
        return False

    def _whole_graphs(self, keep) -> bool:
        """One launch sequence for all encoder blocks (forward) / the whole block chain (backward), captured as ONE hipGraph
        each when graphs are on: only without data parallelism (no collective cuts the sequence) and with every block kept
        (stochastic depth changes the sequence from step to step).  The choice does not depend on whether graphs are enabled,
        so the eager and the replayed path issue exactly the same kernels.  LIDK_WHOLE_GRAPHS=0 restores one sequence per block."""
        return (self._hip and self.stat_allreduce is None and self.on_stage_grads_ready is None
                and self.cfg.n_blocks > 0 and all(keep) and _os_env("LIDK_WHOLE_GRAPHS", "1") == "1")

    def _fused_bn_stats(self) -> bool:
        return self.stat_allreduce is None and hasattr(self.k, "bn_train_stats_from_partials")

    def _run_split(self, key, fn, collective):
        """``fn(part)`` issues a block's launches; they are replayed from captured hipGraphs.  Under data parallelism the
        sequence is cut at the SyncBatchNorm all-reduce (``collective``), which runs eagerly between the two graphs."""
        if collective is None:
            self.graphs.run(key + ("all",), lambda: fn("all"))
        else:
            self.graphs.run(key + ("a",), lambda: fn("a"))
            collective()
            self.graphs.run(key + ("b",), lambda: fn("b"))

    def _bn_collective(self, w: _Work, ci: int, training: bool = True, sums=None):
        if self.stat_allreduce is None or not training:
            return None
        t = (w.sums if sums is None else sums)[:2 * ci + 1]              # sums + row count in one collective
        return lambda: self.stat_allreduce(t)

    def _enc_block_fwd(self, x, i, w: _Work, training: bool, part: str, fuse_next: bool = False, ln_done: bool = False):
        """fuse_next: block i + 1 follows directly, so post_norm and its first PreNorm run as one launch (which leaves h1 /
        mean / rstd of block i + 1 in place); ln_done: this block's first PreNorm was already produced that way."""
        bp, bb = self.enc_params[i], w.enc[i]
        x4 = bb.x4
        # post_norm (and, with fuse_next, the next block's first PreNorm) ride in the second FeedForward's epilogue when it is fused
        tail = dict(gA=bp.post["w"], bA=bp.post["b"], yA32=bb.out, meanA=bb.mean[4], rstdA=bb.rstd[4])
        if fuse_next:
            nb, nbb = self.enc_params[i + 1], w.enc[i + 1]
            tail.update(gB=nb.ff1["ln_w"], bB=nb.ff1["ln_b"], yBT=nbb.h1, meanB=nbb.mean[0], rstdB=nbb.rstd[0])
        done = self._block_fwd(x, bp, bb, w, training, part, ff1_ln_done=ln_done, tail=tail)
        if part in ("all", "b") and not done:
            if fuse_next:
                self.k.layernorm2_fwd(x4, bp.post["w"], bp.post["b"], bb.out, bb.mean[4], bb.rstd[4], nb.ff1["ln_w"],
                                      nb.ff1["ln_b"], nbb.h1, nbb.mean[0], nbb.rstd[0])
            else:
                self.k.layernorm_fwd(x4, bp.post["w"], bp.post["b"], y32=bb.out, mean=bb.mean[4], rstd=bb.rstd[4],
                                     dtype=self.act_dtype)

    def _enc_block_bwd(self, dy, x_in, i, w: _Work, dfeat, part: str, S, wg: bool, post_done: bool = False, fuse=None):
        """post_done: the block that ran before (in backward order) already produced this block's post_norm backward in its
        fused LayerNorm pair; fuse: see _ff_bwd."""
        bp, bb = self.enc_params[i], w.enc[i]
        if part in ("all", "a") and not post_done:
            post = dict(ln_w=bp.post["w"], dln_w=bp.post["dw"], dln_b=bp.post["db"])
            self._ln_bwd(w, dy, bb.x4, bb.mean[4], bb.rstd[4], post, S.lnp[0], wg, dx=w.dxa, dxT=S.dyTs[0], dxT_scale=0.5)
        self._block_bwd(w, x_in, bp, bb, w.dxa, S, dfeat, part, wg, fuse=fuse)

    def _head_fwd(self, w: _Work, feat, lang, training, seed, logits):
        cfg = self.cfg
        bp, bb = self.head_params[lang], w.head
        x4 = bb.x4
        ci = bp.conv["dw"].shape[0]
        drop = training and cfg.dropout > 0
        tail = dict(gA=bp.post["w"], bA=bp.post["b"], meanA=bb.mean[4], rstdA=bb.rstd[4])
        tail["yA32" if drop else "yAT"] = bb.out if drop else w.head_h
        done = []
        self._run_split(("hf", id(w), lang, feat.data_ptr(), training),
                        lambda part: done.append(self._block_fwd(feat, bp, bb, w, training, part, tail=tail)),
                        self._bn_collective(w, ci, training))
        # under graph replay the lambda does not run: whether post_norm rode in the FeedForward epilogue is a property of the shape
        fused_tail = self._ffn_fused(w.M, bb.a4.shape[1]) and _os_env("LIDK_FFN_NEXT_LN", "1") == "1"
        p = f"model.last_projects.{lang}.linear"
        if drop:
            if not fused_tail:
                self.k.layernorm_fwd(x4, bp.post["w"], bp.post["b"], y32=bb.out, mean=bb.mean[4], rstd=bb.rstd[4],
                                     dtype=self.act_dtype)
            self.k.dropout(bb.out, w.head_h, cfg.dropout, seed=seed + 7919, keep_in=self._forced_masks.get("head"),
                        keep_out=w.head_keep)
        elif not fused_tail:
            self.k.layernorm_fwd(x4, bp.post["w"], bp.post["b"], yT=w.head_h, mean=bb.mean[4], rstd=bb.rstd[4])
        self.k.gemm_nt(w.head_h, self.wview(p + ".weight")[0], logits, bias=self.pview(p + ".bias"))

    _forced_masks: Dict[str, torch.Tensor] = {}

    def forward(self, mel: torch.Tensor, lang: Optional[str], training: bool, keep_layers: Optional[List[bool]] = None,
                masks: Optional[Dict[str, torch.Tensor]] = None):
        """mel (B, F, n_mels) f32 on the GPU -> {lang: logits (B, T, V+1) f32}.  In training mode the activations needed
        by ``backward`` stay in the (B, F) workspace until the next forward of the same shape."""
        if not self._built:
            raise LidkError("Engine.forward before Engine.to('cuda')")
        if mel.dtype != torch.float32:
            raise LidkError(f"Engine.forward needs float32 features, got {mel.dtype}")
        if self._hip and not mel.is_cuda:
            raise LidkError(f"Engine.forward got a tensor on {mel.device}: the HIP path has no CPU fallback")
        mel = mel.contiguous()
        B, F_, nm = mel.shape
        want = self.cfg.n_mels if self.cfg.front == "subsample" else self.cfg.d
        if nm != want:
            raise LidkError(f"expected {want} input channels, got {nm}")
        w = self.work(B, F_)
        self._forced_masks = masks or {}
        self.step_count += 1
        seed = ((self.seed + 7919 * self.rank_salt) * 1000003 + self.step_count) & 0x7FFFFFFFFFFF
        x = self._front_fwd(w, mel, training, seed)
        keep = keep_layers if (training and keep_layers is not None) else [True] * self.cfg.n_blocks
        if self._whole_graphs(keep):
            # single process, every block kept: the whole encoder forward is ONE captured sequence (12 x 19 launches, one host
            # call) instead of one per block - the host then issues ~20 calls per step instead of ~60
            x_first = x

            def enc_all():
                xx = x_first
                fuse = self._ln2_ok()
                for i in range(self.cfg.n_blocks):
                    self._enc_block_fwd(xx, i, w, training, "all", fuse_next=fuse and i + 1 < self.cfg.n_blocks,
                                        ln_done=fuse and i > 0)
                    xx = w.enc[i].out

            self.graphs.run(("efA", id(w), x.data_ptr(), training), enc_all)
            x = w.enc[self.cfg.n_blocks - 1].out
        else:
            for i in range(self.cfg.n_blocks):
                if not keep[i]:
                    continue
                self._run_split(("ef", id(w), i, x.data_ptr(), training),
                                lambda part: self._enc_block_fwd(x, i, w, training, part),
                                self._bn_collective(w, self.enc_params[i].conv["dw"].shape[0], training))
                x = w.enc[i].out
        out = {}
        langs = [lang] if lang is not None else list(self.cfg.lang2vocab)
        for l in langs:
            v1 = self.cfg.lang2vocab[l] + 1
            key = (l, B)
            if key not in w.logits:
                w.logits[key] = torch.empty(w.M, v1, device=self.device, dtype=torch.float32)
            self._head_fwd(w, x, l, training, seed, w.logits[key])
            out[l] = w.logits[key].view(B, w.T, v1)
        self._ctx = dict(w=w, feat=x, keep=keep, lang=lang, training=training, front_in=mel)
        self._forced_masks = {}
        if self._hip and training:
            # "the logits are done": the Trainer starts the NEXT batch's feature kernels behind this point, i.e. beside the CTC
            # lattice (64 workgroups, the only sequential stretch of the step) instead of beside the first encoder kernels
            self.fwd_done = torch.cuda.Event()
            self.fwd_done.record()
        return out

    # ------------------------------------------------------------------ CTC loss fused behind the training forward
    def ctc_supported(self) -> bool:
        return bool(self._hip and hasattr(self.k, "ctc_forward") and _os_env("LIDK_CTC_FUSED", "1") == "1")

    def ctc_forward(self, logits: torch.Tensor, texts: torch.Tensor, wav_pct: torch.Tensor, txt_pct: torch.Tensor, blank: int):
        """Mean CTC loss of the last training forward's logits (lid/LidModule_ASR_Supervised.py:162-168): lengths from the batch's
        percents, the three CTC launches and the mean - 5 launches, no torch elementwise kernels; the lattices stay in the
        workspace and ``backward(None, ctc_gscale=...)`` turns them into the vocabulary projection's gradient operand.
        -> (loss 0-dim f32, in_len, tg_len) or None when the lattice does not fit the fast path."""
        w = self._ctx["w"]
        B, T, V1 = logits.shape
        texts = texts.contiguous()
        Lmax = texts.shape[1]
        key = ("ctc", B, T, V1, Lmax)
        c = w.__dict__.setdefault("_ctc_bufs", {}).get(key)
        if c is None:
            dev = logits.device
            c = w._ctc_bufs[key] = dict(ws=torch.empty(max(self.k.ctc_workspace_bytes(B, T, V1, Lmax) // 4, 1), device=dev),
                                        in_len=torch.empty(B, device=dev, dtype=torch.int64),
                                        tg_len=torch.empty(B, device=dev, dtype=torch.int64), per=torch.empty(B, device=dev))
        mean = torch.empty(1, device=logits.device)                # a fresh scalar per step: callers keep the losses they are handed
        if not self.k.ctc_forward(logits, texts, wav_pct.float().contiguous(), txt_pct.float().contiguous(), c["in_len"], c["tg_len"],
                                  c["per"], mean, c["ws"], blank):
            return None
        self._ctc = dict(c, logits=logits, texts=texts, blank=blank, B=B)
        return mean.view(()), c["in_len"], c["tg_len"]

    # ------------------------------------------------------------------ backward pieces
    @staticmethod
    def _splitk(n, k):
        """Split of the M = B*T contraction for a [n, k] weight gradient: enough workgroups to fill 256 CUs, but few enough
        that the float-atomic traffic (splitk * |dW|) stays small next to the operand reads."""
        tiles = -(-n // 64) * -(-k // 64)
        return max(1, min(16, round(_WGRAD_WGS / tiles)))

    def _fork(self):
        """Context for work that may overlap the main stream from here on: the side stream first waits for everything
        issued so far (the producer of the wgrad's dY).  Inside a hipGraph capture this becomes a fork edge."""
        if self.side is None:
            return contextlib.nullcontext()
        self.side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.side)

    def _join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def _wgrad(self, w: _Work, dyT, xT, dW, n, k, db=None):
        """dW [n,k] (f32) += dyT[M,n]^T @ xT[M,k] and db [n] += column sums of dyT, straight from the row-major activations."""
        self.k.gemm_tn(dyT, xT, dW, colsum=db, splitk=self._splitk(n, k), M=w.M, N1=n, N2=k)

    def _ln_bwd(self, w: _Work, dy, x, mean, rstd, P, lnp, wg: bool, **kw):
        """LayerNorm backward.  wg=True finishes dgamma/dbeta at once (shared scratch); otherwise only the partial rows are
        written, into this site's own buffer, and _block_wgrads finishes them beside the next block's chain."""
        if wg:
            self.k.layernorm_bwd(dy, x, mean, rstd, P["ln_w"], w.partial, dgamma=P["dln_w"], dbeta=P["dln_b"],
                                 dtype=self.act_dtype, **kw)
        else:
            w.__dict__.setdefault("_lnp_rows", {})[id(lnp)] = 0        # lidk_layernorm_bwd's own partial-row layout
            self.k.layernorm_bwd(dy, x, mean, rstd, P["ln_w"], lnp, dtype=self.act_dtype, **kw)

    def _dgrad_ln_bwd(self, w: _Work, dy, WT, K: int, x, mean, rstd, P, lnp, wg: bool, dres, dx, dxT, dxT_scale):
        """Data gradient of the projection behind a PreNorm (N = d) followed by that LayerNorm's backward: lidk_gemm_nt +
        lidk_layernorm_bwd through w.dh (LIDK_DGRAD_LN=0), or the one-launch form lidk_dgrad_ln_bwd (csrc/ffn.hip; default).  Before
        the data-gradient chain stayed on one hardware queue (late fork) the fused form measured no faster - 7.43-7.46 vs 7.34-7.43
        ms per step: 151 row-panel workgroups against a 604-workgroup GEMM plus a streaming pass.  Under the late-fork schedule, with
        the chain as the critical path, one launch fewer per site pays: 6.96 / 7.06 against 7.04 / 7.12 (same box, two rounds)."""
        M, d = w.M, self.cfg.d
        key = ("dln", M, K)
        ok = self._split_ok.get(key)
        if ok is None:
            ok = self._split_ok[key] = bool(self._hip and hasattr(self.k, "dgrad_ln_bwd") and _os_env("LIDK_DGRAD_LN", "1") == "1"
                                            and WT.stride(0) == K and self.k.dgrad_ln_bwd_supported(M, d, K, self.act_dtype)
                                            and self.k.ffn_bwd_partial_rows(M) <= L.LN_BWD_BLOCKS)
        lnp_rows = w.__dict__.setdefault("_lnp_rows", {})
        if not ok:
            lnp_rows[id(lnp)] = 0
            self.k.gemm_nt(dy, WT, w.dh, N=d, K=K)
            self._ln_bwd(w, w.dh, x, mean, rstd, P, lnp, wg, dres=dres, dx=dx, dxT=dxT, dxT_scale=dxT_scale)
            return
        rows = self.k.ffn_bwd_partial_rows(M)
        lnp_rows[id(lnp)] = rows
        self.k.dgrad_ln_bwd(dy, WT, x, mean, rstd, P["ln_w"], lnp, dres=dres, dx=dx, dxT=dxT, dxT_scale=dxT_scale)
        if wg:
            self.k.layernorm_param_grads_rows(lnp, rows, d, P["dln_w"], P["dln_b"])

    def _ff_bwd(self, w: _Work, dx_res, dyT, x_in, P, h, a, u, mean, rstd, dx_out, dxT_out, dxT_scale, da_buf, wg: bool, lnp,
                fuse=None):
        """dyT = 0.5*dx_res (T).  Produces dx_out = dx_res + LN'(dh) and optional T copy for the next stage.
        fuse = (bb_prev, post_prev, S_prev, forked): x_in is the output of the encoder block that comes NEXT in backward order, so this
        PreNorm's backward and that block's post_norm backward run as one launch: dx (f32) -> w.dxa, 0.5*dx (T) ->
        S_prev.dyTs[0], partial rows of both LayerNorms into their own buffers; dx_out is not written."""
        M, d, ff = w.M, self.cfg.d, a.shape[1]
        lnp_rows = w.__dict__.setdefault("_lnp_rows", {})       # partial-row count of the sites whose rows came from lidk_ffn_bwd
        if wg:
            self._wgrad(w, dyT, u, P["dw2"], d, ff, P["db2"])
        da = da_buf if da_buf.shape[1] == ff else da_buf.view(-1)[:M * ff].view(M, ff)
        if self._ffn_fused(M, ff):
            # both data-gradient products (and, unless this PreNorm's backward is paired with the next block's post_norm, the
            # LayerNorm backward too) in one launch; the partial (dgamma | dbeta) rows are one per 64-row workgroup
            rows = self.k.ffn_bwd_partial_rows(M)
            # LIDK_FFN_LN_PAIR=1: both LayerNorm backwards of the block boundary in the kernel's epilogue (lidk_ffn_bwd_ln2).  Correct
            # (tests/test_gpu_ffn.py) but measured slower end to end - 7.44 / 7.44 vs 7.38 / 7.38 ms per step: the long epilogue of 151
            # lock-step workgroups costs more than the 11 us streaming launch it removes - so the dh form + lidk_layernorm2_bwd stay
            # (re-measured under the late-fork schedule: 7.19 / 7.19 vs 7.14 / 7.16).
            pair_fused = fuse is not None and _os_env("LIDK_FFN_LN_PAIR", "0") == "1"
            lnp_rows[id(lnp)] = rows if (fuse is None or pair_fused) else 0
            if pair_fused:
                # this PreNorm's backward AND the following block's post_norm backward in the kernel's epilogue (was: dh out +
                # lidk_layernorm2_bwd): dx (f32) -> w.dxa, 0.5 * dx (T) -> S_prev.dyTs[0], partial rows of both LayerNorms
                pbb, post, Sp, forked = fuse
                if forked:          # S_prev's buffers are still read by the weight gradients running on the second stream
                    self._join()
                lnp_rows[id(Sp.lnp[0])] = rows
                self.k.ffn_bwd(dyT, a, P["w1"][1], P["w2"][1], da, x=x_in, mean=mean, rstd=rstd, gamma=P["ln_w"], dres=dx_res,
                               dx=w.dxa, dxT=Sp.dyTs[0], dxT_scale=0.5, partial=lnp,
                               pair=dict(x1=pbb.x4, mean1=pbb.mean[4], rstd1=pbb.rstd[4], gamma1=post["w"], partial1=Sp.lnp[0]))
                if wg:
                    self._wgrad(w, da, h, P["dw1"], ff, d, P["db1"])
                return
            if fuse is None:
                self.k.ffn_bwd(dyT, a, P["w1"][1], P["w2"][1], da, x=x_in, mean=mean, rstd=rstd, gamma=P["ln_w"], dres=dx_res,
                               dx=dx_out, dxT=dxT_out, dxT_scale=dxT_scale, partial=lnp)
                if wg:
                    self._wgrad(w, da, h, P["dw1"], ff, d, P["db1"])
                    self.k.layernorm_param_grads_rows(lnp, rows, d, P["dln_w"], P["dln_b"])
                return
            self.k.ffn_bwd(dyT, a, P["w1"][1], P["w2"][1], da, dh=w.dh)
            if wg:
                self._wgrad(w, da, h, P["dw1"], ff, d, P["db1"])
        else:
            self.k.gemm_nt(dyT, P["w2"][1], da, act=L.ACT_SWISH_GRAD, aux=a, N=ff, K=d)
            if wg:
                self._wgrad(w, da, h, P["dw1"], ff, d, P["db1"])
            self.k.gemm_nt(da, P["w1"][1], w.dh, N=d, K=ff)
        if fuse is not None:
            pbb, post, Sp, forked = fuse
            if forked:          # S_prev's buffers are still read by the weight gradients running on the second stream
                self._join()
            self.k.layernorm2_bwd(w.dh, dx_res, x_in, mean, rstd, P["ln_w"], pbb.x4, pbb.mean[4], pbb.rstd[4], post["w"],
                                  w.dxa, Sp.dyTs[0], 0.5, Sp.lnp[0], lnp)
            return
        self._ln_bwd(w, w.dh, x_in, mean, rstd, P, lnp, wg, dres=dx_res, dx=dx_out, dxT=dxT_out, dxT_scale=dxT_scale)

    def _block_bwd(self, w: _Work, x_in, bp: _BlockParams, bb: _BlockBuf, dx4, S, dx_in_out, part: str = "all",
                   wg: bool = True, fuse=None):
        """dx4: f32 gradient at x4 (after post_norm backward); S.dyTs[0] = 0.5*dx4 in T.  Writes the gradient w.r.t. the
        block input into dx_in_out (f32) and every weight-gradient operand (dY) into the scratch set S.  ``part`` cuts the
        sequence at the SyncBatchNorm backward all-reduce.  wg=False leaves the weight gradients to _block_wgrads."""
        B, T, M, d = w.B, w.T, w.M, self.cfg.d
        a, b = (w.dxa, w.dxb) if dx4 is w.dxb else (w.dxb, w.dxa)     # two f32 ping-pong buffers
        C = bp.conv
        ci, K = C["dw"].shape[0], C["dw"].shape[2]
        pad_left = K // 2
        ds = S.ds.view(-1)[:M * ci].view(M, ci)
        dx3 = a
        t0, t1, t2, t3 = S.dyTs
        if part in ("all", "a"):
            # ---- ff2: y = x3 + 0.5*ff(x3)
            self._ff_bwd(w, dx4, t0, bb.x3, bp.ff2, bb.h4, bb.a4, bb.u4, bb.mean[3], bb.rstd[3], a, t1, 1.0, S.da[0], wg, S.lnp[1])
            if self._after_first is not None:           # whole-chain capture: the previous block's weight gradients fork here
                cb, self._after_first = self._after_first, None
                cb()
            # ---- conv module: y = x2 + conv(x2)
            if wg:
                self._wgrad(w, t1, bb.s, C["dw2"].view(d, ci), d, ci, C["db2"])
            # data gradient of the second pointwise convolution; its epilogue also leaves the BatchNorm + Swish backward sums of
            # the tile (lidk_gemm_nt_bn_sums: one launch and one pass over ds less on the data-gradient chain, LIDK_BN_GEMM=0 or a
            # shape outside the pipelined kernel: GEMM, then the streaming reduction)
            nparts = 0
            if self._hip and hasattr(self.k, "gemm_nt_bn_sums") and _os_env("LIDK_BN_GEMM", "1") == "1" \
                    and (M // 64) * 2 * 2 * ci <= w.partial.numel():
                nparts = self.k.gemm_nt_bn_sums(t1, C["w2"][1], ds, bb.c, bb.bn_mean, bb.bn_rstd, C["bn_w"], C["bn_b"], w.partial,
                                                M=M, N=ci, K=d)
            if not nparts:
                self.k.gemm_nt(t1, C["w2"][1], ds, N=ci, K=d)
                self.k.bn_swish_bwd_reduce(ds, bb.c, bb.bn_mean, bb.bn_rstd, C["bn_w"], C["bn_b"], w.partial)
                nparts = L.BN_PARTIAL_BLOCKS
            # sums is all-reduced in place by the SyncBN collective under DP; sums_local keeps this rank's share
            self.k.reduce_partials_f64(w.partial, nparts, 2 * ci, S.sums[:2 * ci + 1], S.sums_local[:2 * ci + 1], tail=M)
        if part in ("all", "b"):
            # BatchNorm+Swish backward, depthwise-conv input gradient and GLU backward in one launch; the conv's weight
            # gradient (which needs dc materialised) goes with the other weight gradients
            dy1 = S.dy1.view(-1)[:M * 2 * ci].view(M, 2 * ci)
            self.k.dwconv_bwd_input_bn_glu(ds, bb.c, bb.bn_mean, bb.bn_rstd, C["bn_w"], C["bn_b"], S.sums[:2 * ci + 1],
                                           0, C["dw"].view(ci, K), bb.y, dy1, B, T, pad_left)
            if wg:
                self._conv_wgrad(w, bp, bb, S)
                self._wgrad(w, dy1, bb.h3, C["dw1"].view(2 * ci, d), 2 * ci, d, C["db1"])
            self._dgrad_ln_bwd(w, dy1, C["w1"][1], 2 * ci, bb.x2, bb.mean[2], bb.rstd[2], C, S.lnp[2], wg, dx3, b, t2, 1.0)
            dx2 = b
            # ---- attention: y = x1 + attn(x1)
            A = bp.attn
            inner = bp.heads * bp.dh
            if wg:
                self._wgrad(w, t2, bb.o, A["dwo"], d, inner, A["dbo"])
            do = w.dmid.view(-1)[:M * inner].view(M, inner)
            self.k.gemm_nt(t2, A["wo"][1], do, N=inner, K=d)
            dqkv = S.dqkv.view(-1)[:M * 3 * inner].view(M, 3 * inner)
            # deferred mode: the relative-position embedding's gradient (a weight gradient, a third of this step's work)
            # is left to _block_wgrads, which computes it from the dS rows kept in the scratch set
            split = (not wg) and self._relpos_split(T, bp.dh)
            self.k.attn_bwd(bb.qkv, A["emb"], None if self._attn_recompute(T, bp.dh) else bb.probs, do, dqkv,
                            None if split else A["demb"], S.dsc, B, T, bp.heads, bp.dh, rel_emb_T=A["embT"])
            if wg:
                self._wgrad(w, dqkv, bb.h2, A["dwqkv"], 3 * inner, d)
            self._dgrad_ln_bwd(w, dqkv, A["wqkv"][1], 3 * inner, bb.x1, bb.mean[1], bb.rstd[1], A, S.lnp[3], wg, dx2, a, t3, 0.5)
            dx1 = a
            # ---- ff1
            self._ff_bwd(w, dx1, t3, x_in, bp.ff1, bb.h1, bb.a1, bb.u1, bb.mean[0], bb.rstd[0], dx_in_out, None, 1.0, S.da[1], wg,
                         S.lnp[4], fuse=fuse)

    def _conv_wgrad(self, w: _Work, bp: _BlockParams, bb: _BlockBuf, S):
        """BatchNorm parameter gradients and the depthwise-conv weight gradient: dc is materialised here (off the dgrad
        chain, which forms it on the fly inside dwconv_bwd_input_bn_glu)."""
        B, T, M = w.B, w.T, w.M
        C = bp.conv
        ci, K = C["dw"].shape[0], C["dw"].shape[2]
        ds = S.ds.view(-1)[:M * ci].view(M, ci)
        if self._hip and self.k.dwconv_bwd_weight_bn_supported(ci, self.act_dtype):     # one pass, dc never written
            self.k.dwconv_bwd_weight_bn(ds, bb.c, bb.bn_mean, bb.bn_rstd, C["bn_w"], C["bn_b"], S.sums[:2 * ci + 1],
                                        S.sums_local[:2 * ci + 1], 0, bb.g, C["ddw"].view(ci, K), C["ddwb"], C["dbn_w"],
                                        C["dbn_b"], w.dw_partial, B, T, K // 2)
            return
        dc = S.dc.view(-1)[:M * ci].view(M, ci)
        self.k.bn_swish_bwd_apply(ds, bb.c, bb.bn_mean, bb.bn_rstd, C["bn_w"], C["bn_b"], S.sums[:2 * ci + 1],
                                  S.sums_local[:2 * ci + 1], 0, dc, C["dbn_w"], C["dbn_b"])
        self.k.dwconv_bwd_weight(dc, bb.g, C["ddw"].view(ci, K), C["ddwb"], w.dw_partial, B, T, K // 2)

    def _block_wgrads(self, w: _Work, bp: _BlockParams, bb: _BlockBuf, S, post_norm: bool):
        """The ten weight-gradient launches of one block, reading the dY operands its dgrad chain left in scratch set S, and the
        LayerNorm dgamma/dbeta finalizers of its sites (post_norm: the block's own post_norm went through this path too)."""
        B, T, M, d = w.B, w.T, w.M, self.cfg.d
        sites = [(1, bp.ff2), (2, bp.conv), (3, bp.attn), (4, bp.ff1)]
        if post_norm:
            sites.append((0, dict(dln_w=bp.post["dw"], dln_b=bp.post["db"])))
        lnp_rows = w.__dict__.get("_lnp_rows", {})
        # The block's five finalisers as one launch (LIDK_LN_GROUPED=0: five launches).  With the register-staged weight gradients
        # this measured SLOWER (7.78-7.84 vs 7.68-7.69 ms per step: the weight-gradient stream reached its big grouped GEMM earlier
        # and took more of the chip from the data-gradient chain at the start of the block); with the LDS-DMA weight gradients and
        # the late fork it is the faster form: 7.00 / 7.08 against 7.07 / 7.15 (same box, two rounds).
        if hasattr(self.k, "layernorm_param_grads_grouped") and _os_env("LIDK_LN_GROUPED", "1") == "1":
            rows_of = [lnp_rows.get(id(S.lnp[i]), 0) or self.k.layernorm_bwd_partial_rows(M) for i, _ in sites]
            cache = w.__dict__.setdefault("_ln_groups", {})
            key = (id(bp), id(S), post_norm, tuple(rows_of))
            grp = cache.get(key)
            if grp is None:
                grp = cache[key] = self.k.build_ln_param_group([(S.lnp[i], r, d, P["dln_w"], P["dln_b"])
                                                                for (i, P), r in zip(sites, rows_of)])
            self.k.layernorm_param_grads_grouped(grp)
        else:
            for i, P in sites:
                rows = lnp_rows.get(id(S.lnp[i]), 0)
                if rows:
                    self.k.layernorm_param_grads_rows(S.lnp[i], rows, d, P["dln_w"], P["dln_b"])
                else:
                    self.k.layernorm_param_grads(S.lnp[i], M, d, P["dln_w"], P["dln_b"])
        C, A = bp.conv, bp.attn
        ci, K = C["dw"].shape[0], C["dw"].shape[2]
        inner = bp.heads * bp.dh
        t0, t1, t2, t3 = S.dyTs
        dy1 = S.dy1.view(-1)[:M * 2 * ci].view(M, 2 * ci)
        dqkv = S.dqkv.view(-1)[:M * 3 * inner].view(M, 3 * inner)
        sites = []                                   # (dY, activation, dW, db, n, k): dW [n, k] += dY^T @ activation
        for P, dyT, da_buf, h, u in ((bp.ff2, t0, S.da[0], bb.h4, bb.u4), (bp.ff1, t3, S.da[1], bb.h1, bb.u1)):
            ff = u.shape[1]
            da = da_buf if da_buf.shape[1] == ff else da_buf.view(-1)[:M * ff].view(M, ff)
            sites += [(dyT, u, P["dw2"], P["db2"], d, ff), (da, h, P["dw1"], P["db1"], ff, d)]
        sites += [(t1, bb.s, C["dw2"].view(d, ci), C["db2"], d, ci), (dy1, bb.h3, C["dw1"].view(2 * ci, d), C["db1"], 2 * ci, d),
                  (t2, bb.o, A["dwo"], A["dbo"], d, inner), (dqkv, bb.h2, A["dwqkv"], None, 3 * inner, d)]
        if self._grouped_wgrads():
            # ONE launch for the block's eight Linear / 1x1-conv weight gradients (csrc/gemm.hip gemm_tn_grouped_kernel); the
            # descriptor table holds raw pointers into this workspace and the arenas, so it lives and dies with the workspace
            cache = w.__dict__.setdefault("_tn_groups", {})
            key = (id(bp), id(bb), id(S))
            grp = cache.get(key)
            if grp is None:
                tile = _WGRAD_TILE if all(n % 128 == 0 and k % 128 == 0 for _, _, _, _, n, k in sites) and M % 64 == 0 else 64
                split = _WGRAD_SPLIT if tile == _WGRAD_TILE else 4
                grp = cache[key] = self.k.build_tn_group([(dy, x, dW, db, M, n, k) for dy, x, dW, db, n, k in sites],
                                                         split=split, tile=tile)
            self.k.gemm_tn_grouped(grp)
        else:
            for dy, x, dW, db, n, k in sites:
                self._wgrad(w, dy, x, dW, n, k, db)
        self._conv_wgrad(w, bp, bb, S)
        if self._relpos_split(T, bp.dh):
            self.k.attn_bwd_relpos(bb.qkv, S.dsc, bb.ldp, A["demb"], B, T, bp.heads, bp.dh)

    def _attn_recompute(self, T: int, dh: int) -> bool:
        """No probabilities in HBM: the attention backward recomputes them (bf16 MFMA shapes; LIDK_ATTN_RECOMPUTE=0 turns it off)."""
        key = ("rc", T, dh)
        if key not in self._split_ok:
            self._split_ok[key] = bool(self._hip and hasattr(self.k, "attn_recompute_supported")
                                       and self.k.attn_recompute_supported(T, dh, self.act_dtype))
        return self._split_ok[key]

    def _grouped_wgrads(self) -> bool:
        """The block's weight-gradient GEMMs as one grouped launch (bf16 operands, HIP backend; LIDK_WGRAD_GROUPED=0 restores the
        eight separate launches)."""
        return bool(self._hip and self.act_dtype == torch.bfloat16 and hasattr(self.k, "gemm_tn_grouped")
                    and _os_env("LIDK_WGRAD_GROUPED", "1") == "1")

    def _relpos_split(self, T: int, dh: int) -> bool:
        key = (T, dh)
        if key not in self._split_ok:
            self._split_ok[key] = bool(self._hip and hasattr(self.k, "attn_bwd_relpos_supported")
                                       and self.k.attn_bwd_relpos_supported(T, dh, self.act_dtype))
        return self._split_ok[key]

    def _backward_blocks(self, w: _Work, blocks, dfeat, defer: bool):
        """One captured sequence per block (cut at the SyncBatchNorm all-reduce under data parallelism)."""
        prev = None                                  # (tag, bp, bb, S, stage)
        for n, (kind, tag, bpk, bbk, x_in, stage) in enumerate(blocks):
            S = w.sets[n & 1]

            def fn(part, kind=kind, tag=tag, bpk=bpk, bbk=bbk, x_in=x_in, S=S, prev=prev):
                if prev is not None and part in ("all", "a"):
                    with self._fork():
                        self._block_wgrads(w, prev[1], prev[2], prev[3], prev[0][0] == "enc")
                if kind == "head":
                    self._block_bwd(w, x_in, bpk, bbk, w.dxa, S, dfeat, part, not defer)
                else:
                    self._enc_block_bwd(dfeat, x_in, tag, w, dfeat, part, S, not defer)
                if prev is not None and part in ("all", "a"):
                    self._join()

            key = ("bb", id(w), kind, tag, x_in.data_ptr(), n & 1, prev[0] if prev else None)
            self._run_split(key, fn, self._bn_collective(w, bpk.conv["dw"].shape[0], sums=S.sums))
            if defer:
                if prev is not None and self.on_stage_grads_ready:
                    self.on_stage_grads_ready(prev[4])
                prev = ((kind, tag), bpk, bbk, S, stage)
            elif self.on_stage_grads_ready:
                self.on_stage_grads_ready(stage)
        if prev is not None:
            self._block_wgrads(w, prev[1], prev[2], prev[3], prev[0][0] == "enc")
            if self.on_stage_grads_ready:
                self.on_stage_grads_ready(prev[4])

    def backward(self, dlogits: Optional[torch.Tensor], ctc_gscale: Optional[torch.Tensor] = None):
        """dlogits (B, T, V+1) f32 for the language of the last training forward.  Accumulates into ``grad``.
        ctc_gscale (with dlogits None): the upstream gradient of the mean loss ``ctc_forward`` returned - a device scalar; the CTC
        gradient kernel then writes the operand of the vocabulary projection's gradient GEMMs itself (T-typed, padded columns
        zero, scaled by ctc_gscale / B): no f32 dlogits tensor, no converting copy."""
        ctx = self._ctx
        if not ctx["training"] or ctx["lang"] is None:
            raise LidkError("Engine.backward needs a preceding training-mode forward with a single language")
        cfg, w, lang = self.cfg, ctx["w"], ctx["lang"]
        M, d = w.M, cfg.d
        v1 = cfg.lang2vocab[lang] + 1
        v1p = _ceil(v1, 8)
        p = f"model.last_projects.{lang}.linear"
        dlT = w.dlT.view(-1)[:M * v1p].view(M, v1p)
        if dlogits is None:
            c = getattr(self, "_ctc", None)
            if c is None or ctc_gscale is None or c["logits"].shape[-1] != v1:
                raise LidkError("Engine.backward(None): needs a preceding ctc_forward of the same step and its upstream gradient")
            self.k.ctc_backward(c["logits"], c["texts"], c["in_len"], c["tg_len"], dlT, c["ws"], c["blank"], 1.0 / c["B"],
                                ctc_gscale.reshape(1).float())
            self._ctc = None
        else:
            dl = dlogits.contiguous().view(M, v1)
            if v1p != v1:
                dlT.zero_()
            self.k.scale_cast_2d(dl, dlT, M, v1)
        bp, bb = self.head_params[lang], w.head

        def head_prologue():
            # vocabulary projection, head dropout, the head block's post_norm: fixed buffers, no per-step scalars
            self._wgrad(w, dlT, w.head_h, self.gview(p + ".weight"), v1, d, self.gview(p + ".bias"))
            self.k.gemm_nt(dlT, self.wview(p + ".weight")[1][:, :v1p], w.dh, N=d, K=v1p)
            dy_ln = w.dh
            if cfg.dropout > 0:
                self.k.dropout(w.dh, w.dyT, cfg.dropout, keep_in=w.head_keep)
                dy_ln = w.dyT
                # post_norm backward consumes dy_ln; it must not alias its dxT output
                self.k.scale_cast(dy_ln, w.dh, 1.0)
                dy_ln = w.dh
            self.k.layernorm_bwd(dy_ln, bb.x4, bb.mean[4], bb.rstd[4], bp.post["w"], w.partial, dx=w.dxa, dxT=w.dyT,
                                 dxT_scale=0.5, dgamma=bp.post["dw"], dbeta=bp.post["db"], dtype=self.act_dtype)

        def front_epilogue():
            # front end: x0 = sqrt(d) * (r @ Wl^T + b) [dropout]; r = relu(col @ Wc^T + bc)
            fz = "model.featurizer.sub_sampling"
            dy = w.dfeat                                 # f32 gradient at the first block's input (after pos-enc dropout)
            if cfg.pos_dropout > 0:
                self.k.dropout(dy, w.dxa, cfg.pos_dropout, keep_in=w.pos_keep)
                dy = w.dxa
            self.k.scale_cast(dy, w.dyT, math.sqrt(d))
            nm = cfg.n_mels
            self._wgrad(w, w.dyT, w.r, self.gview(fz + ".linear.weight"), d, nm, self.gview(fz + ".linear.bias"))
            dr = w.dmid.view(-1)[:M * nm].view(M, nm)
            self.k.gemm_nt(w.dyT, self.wview(fz + ".linear.weight")[1], dr, N=nm, K=d)
            self.k.relu_bwd(dr, w.r, dr)
            w.dconv3.zero_()
            self._wgrad(w, dr, w.col, w.dconv3, nm, 3 * nm, self.gview(fz + ".sub_sampling.0.bias"))
            self.gview(fz + ".sub_sampling.0.weight").add_(w.dconv3.view(nm, 3, nm).permute(0, 2, 1))   # layout glue

        whole = self._whole_graphs(ctx["keep"])
        if not whole:
            head_prologue()
        dfeat = w.dfeat
        feat = ctx["feat"]
        # Blocks in backward order: the head block, then the kept encoder blocks in reverse.  In deferred mode (side stream)
        # a block's graph holds its dgrad chain on the main stream and, forked beside it, the weight gradients of the
        # PREVIOUS block (whose dY operands sit in the other scratch set); the last block's weight gradients run at the end.
        defer = self.side is not None
        kept = [i for i in range(cfg.n_blocks) if ctx["keep"][i]]
        blocks = [("head", lang, bp, bb, feat, f"head.{lang}")]
        for idx in reversed(range(len(kept))):
            i = kept[idx]
            x_in = w.enc[kept[idx - 1]].out if idx > 0 else (w.x0d if (cfg.pos_dropout > 0) else w.x0)
            blocks.append(("enc", i, self.enc_params[i], w.enc[i], x_in, f"enc.{i}"))
        if whole:
            # the whole chain - the head's prologue (vocabulary projection, dropout, post_norm), every block's dgrad sequence with
            # the previous block's weight gradients forked beside it, the last block's weight gradients and the front end's
            # backward - as one captured sequence per (workspace, language): between the CTC gradient kernel and the optimizer the
            # host issues ONE call
            poison = _os_env("LIDK_POISON_SCRATCH", "0") == "1"

            def bwd_all():
                head_prologue()
                prv = None
                self._after_first = None              # (a capture that raised half-way must not leave its callback behind)
                fuse_ok = defer and self._ln2_ok()
                post_done = False
                for n, (kind, tag, bpk, bbk, x_in, stage) in enumerate(blocks):
                    ns = len(w.sets)
                    S = w.sets[n % ns]
                    if poison:
                        # Debug mode that pins the schedule's invariant (tests/test_gpu_parity_r2.py): a block may only write its
                        # scratch set once nobody - in particular the weight-gradient stream, still busy with the block that used
                        # the set before - reads it any more.  Everything in the set except what the previous block's tail has
                        # already left there for this block (dyTs[0], the post_norm partial rows lnp[0]) is overwritten with NaN at
                        # the earliest moment the chain could touch it: a missing join turns into NaN gradients, not into a 3e-4
                        # drift that only a lucky comparison notices.
                        for name_, v_ in vars(S).items():
                            for j_, t_ in enumerate(v_ if isinstance(v_, list) else [v_]):
                                if torch.is_tensor(t_) and t_.is_floating_point() and not (name_ in ("dyTs", "lnp") and j_ == 0):
                                    t_.fill_(float("nan"))
                    # x_in of this block is the output of the block that follows in backward order (an encoder block, unless
                    # this is encoder block 0): fuse this block's first-PreNorm backward with that block's post_norm backward
                    fuse = None
                    if fuse_ok and n + 1 < len(blocks):
                        nxt = blocks[n + 1]
                        fuse = (nxt[3], nxt[2].post, w.sets[(n + 1) % ns], prv is not None and ns < 3)
                    late = prv is not None and _FORK_LATE
                    if late:
                        # same dependencies as the fork below, but captured AFTER the block's first data-gradient launch: that
                        # kernel is then the first child of the previous block's last kernel in the hipGraph (and BEFORE the
                        # rest of the chain, whose fused LayerNorm pair joins the second stream ahead of overwriting S_prev)
                        fork_at = torch.cuda.Event()
                        fork_at.record()

                        def side_work(prv=prv, fork_at=fork_at):
                            self.side.wait_event(fork_at)
                            with torch.cuda.stream(self.side):
                                self._block_wgrads(w, prv[1], prv[2], prv[3], prv[0][0] == "enc")
                        self._after_first = side_work
                    elif prv is not None:
                        with self._fork():
                            self._block_wgrads(w, prv[1], prv[2], prv[3], prv[0][0] == "enc")
                    if kind == "head":
                        self._block_bwd(w, x_in, bpk, bbk, w.dxa, S, dfeat, "all", not defer, fuse=fuse)
                    else:
                        self._enc_block_bwd(dfeat, x_in, tag, w, dfeat, "all", S, not defer, post_done=post_done, fuse=fuse)
                    post_done = fuse is not None
                    if late and self._after_first is not None:      # not consumed by the chain: issue it now
                        cb, self._after_first = self._after_first, None
                        cb()
                    if prv is not None:
                        self._join()
                    if defer:
                        prv = ((kind, tag), bpk, bbk, S, stage)
                if prv is not None:
                    self._block_wgrads(w, prv[1], prv[2], prv[3], prv[0][0] == "enc")
                if cfg.front == "subsample":
                    front_epilogue()

            self.graphs.run(("bbA", id(w), lang, feat.data_ptr(), poison), bwd_all)
        else:
            self._backward_blocks(w, blocks, dfeat, defer)
        if cfg.front != "subsample":                 # backbone features: hand d(loss)/d(features) to the caller
            return dfeat[:M].view(ctx["front_in"].shape)
        if not whole:
            front_epilogue()
        if self.on_stage_grads_ready:
            self.on_stage_grads_ready("front")

    # ------------------------------------------------------------------ bookkeeping for the optimizer / DDP
    def active_tensor_ids(self, lang: str, keep: List[bool]) -> List[int]:
        """Tensors that receive a gradient in a step (everything else has 'grad None')."""
        ids = []
        a, b = self.stages["front"]
        ids += [t for t in range(a, b) if not self.specs[t].name.startswith("model.featurizer.linear.")]
        for i in range(self.cfg.n_blocks):
            if keep[i]:
                a, b = self.stages[f"enc.{i}"]
                ids += list(range(a, b))
        a, b = self.stages[f"head.{lang}"]
        ids += list(range(a, b))
        return ids

    def single_active_head(self, tids) -> Optional[str]:
        """The one language whose head holds gradients among tensor ids ``tids``; None if several (or no) heads do."""
        langs = []
        for l in self.cfg.lang2vocab:
            a, b = self.stages[f"head.{l}"]
            if any(a <= t < b for t in tids):
                langs.append(l)
        return langs[0] if len(langs) == 1 else None

    def stage_range(self, stage: str):
        a, b = self.stages[stage]
        if a == b:
            return 0, 0
        lo = self.specs[a].offset
        hi = self.specs[b - 1].offset + _ceil(self.specs[b - 1].numel, ALIGN)
        return lo, hi
