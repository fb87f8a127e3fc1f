"""Training step and loop of the hot path (reference train.py).

``build_graph`` (train.py:74-174) assembled model + loss + optimizer into one TF
graph executed by ``sess.run([train_op, global_step, loss])`` (train.py:317).
Here ``TrainStep`` owns the device buffers and enqueues the same work as HIP
kernels on one stream:

    sample + gather + l2norm  ->  FC1  ->  FC2  ->  l2norm  ->  hinge loss + dE
    ->  l2norm/lrelu bwd  ->  dW2,db2  ->  dH1  ->  dW1,db1
    ->  [all-reduce grads]  ->  Adam | LARS  ->  step counter += 1

Nothing in ``step()`` allocates or synchronises, so the whole step can be
captured into a hipGraph (``use_graph=True``); the step counter and learning
rate live in device memory so replays advance the sampler and Adam's bias
correction.  ``Trainer`` mirrors the reference's loop (train.py:260-336).
"""
import logging
import os
import time

import torch

from . import engine, engine_bf16, engine_f16x2, engine_x3, ops
from .inputs import MODE_INBATCH, MODE_UNIFORM

# sampler mode per negative policy: "semihard" samples like "inbatch" (rows a_i, p_i)
# and mines the negative among the embedded rows of the batch (BASELINE config 2)
_MODES = {"uniform": MODE_UNIFORM, "inbatch": MODE_INBATCH, "semihard": MODE_INBATCH}


def exponential_decay(base_lr, global_step, decay_steps, decay_rate, staircase=True):
    """tf.train.exponential_decay (train.py:108-113)."""
    p = global_step / float(decay_steps)
    if staircase:
        p = float(int(p))
    return base_lr * (decay_rate ** p)


class TrainStep:
    GRAD_SYNC_MODES = ("bucketed", "two", "single")

    def __init__(self, table, pairs, batch_size, feature_size=None, output_size=256,
                 hidden_size=5000, margin=0.8, mode="uniform", optimizer="adam",
                 base_learning_rate=0.01, learning_rate_decay_examples=1000000,
                 learning_rate_decay=0.96, seed=1234, weight_seed=42, device="cuda:0",
                 exchange=None, grad_sync=None, slot0=0, batch_global=None, use_graph=False,
                 prefetch=True, precision="auto", train_table=False, gather_ahead="auto",
                 clip_gradient_norm=0.0, regularization_penalty=0.0, l2_penalty=1e-8,
                 grad_sync_mode="bucketed"):
        """table: FeatureTable (whole catalogue, or this rank's shard when
        ``exchange`` is given); pairs: int32 [P,2] device tensor; ``exchange`` /
        ``grad_sync``: the multi-GPU hooks of cdml_amd.dist (None on one GPU).
        ``train_table``: also train the catalogue rows (lazy Adam, states stored beside the
        shard; build-defined -- the reference keeps the features frozen, train.py:265).
        ``precision``: "auto" (default) = "bf16" for an fp16 catalogue (BASELINE config 4), else "f32x3" -- fp32 operands as
        three exact bf16 planes, six plane products per fp32 product on the bf16 MFMA, held to the fp32 path's bounds; what
        bench.py times -- when the step's rows are a multiple of 128, else "f32" (the fp32 MFMA).  "f16x2" (never chosen by
        "auto"): fp32 operands as TWO fp16 planes under per-tensor power-of-two scales, three plane products on the fp16 MFMA
        -- half of "f32x3"'s matrix work at the same error bound, an fp32 equivalent while its delayed scales hold
        (engine_f16x2.py); under ``use_graph`` (one GPU) its check steps run eagerly and a moved scale re-captures.
        ``use_graph``: False = eager; True = the whole step replayed from ONE hipGraph (single GPU: the
        fast form; data-parallel: the exchange of step t+1 is then recorded on the capturing stream,
        ahead of the forward pass, not under it); "split" (data-parallel with the prefetcher) = three
        graphs per step -- the exchange captured with the prefetch stream as capture origin, forward +
        loss and backward + all-reduce + optimizer with the compute stream as origin -- replayed on
        their own streams and ordered by events recorded eagerly between the replays, so the exchange
        of step t+1 runs UNDER step t's forward GEMMs as in the eager step and every RCCL stream is one
        fork from its capture origin (two deep hangs on this stack: dist.Prefetcher).
        ``gather_ahead``: steps fetched per launch of the fused sampler+gather (single-GPU
        path): the sampler is counter-based, so one launch samples and gathers the rows of
        this step and the next gather_ahead-1 into their own buffers.  "auto" (default): as many
        steps (1 .. 4) as keep the bytes one launch WRITES near the 256 MB Infinity Cache -- more
        steps amortise the launch's fixed cost, but rows written past what the cache absorbs
        cost HBM bandwidth twice (round 5, profiles/r05_gather_sweep.txt: three-plane rows at
        16 384 rows a step read+write 0.70-0.72 of 8 TB/s at two steps per launch, 0.63 at four).
        ``clip_gradient_norm`` / ``regularization_penalty``: build_graph's switches
        (train.py:133-145; the reference's run passes 0 for both, train.py:221-222): per-variable
        tf.clip_by_norm, and penalty * sum_W l2_penalty*|W|^2/2 added to the loss (models.py:28).
        ``optimizer``: "adam" (build_graph's default), "lars" (what the reference's main() uses,
        train.py:354) or "momentum" (Nesterov, momentum 0.9: train.py:115-116).
        ``grad_sync_mode`` (with ``grad_sync``): "bucketed" = the weight gradients as split-K launches
        in buckets whose all-reduce starts as soon as each is enqueued (hidden under the GEMMs that
        follow); "two" = the first layer's weight gradient as ONE full split-K launch, its all-reduce
        (86 % of the bytes) under the second layer's launch, [dW2|db2] after it; "single" = the one
        stream-K launch of the single-GPU step followed by ONE all-reduce of the whole flat gradient
        (fastest kernels, the collective exposed)."""
        if mode not in _MODES:
            raise ValueError("mode must be 'uniform', 'inbatch' or 'semihard'")
        if optimizer not in ("adam", "lars", "momentum"):
            raise ValueError("optimizer must be 'adam', 'lars' or 'momentum'")
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is not None \
                and torch.cuda.current_device() != self.device.index:
            # the ops launch on the CURRENT device's stream: a step built for another device
            # would run its kernels on the wrong stream with that device's pointers
            raise ValueError("TrainStep(device=%s) needs torch.cuda.set_device(%d) first (current device %d)"
                             % (self.device, self.device.index, torch.cuda.current_device()))
        self.table, self.pairs = table, pairs
        # the reference raises IndexError on an id outside the table (inputs.py:158); the fused
        # kernel clamps, so the ids are checked once here instead of per step
        if pairs.numel():
            lo, hi = int(pairs.min().item()), int(pairs.max().item())
            if lo < 0 or hi >= table.n_rows_global:
                raise IndexError("co-watch pair ids span [%d, %d] but the catalogue has %d rows"
                                 % (lo, hi, table.n_rows_global))
        self.B = int(batch_size)
        self.mode = mode
        self.rows_per_triplet = 3 if mode == "uniform" else 2
        self.R = self.B * self.rows_per_triplet
        self.margin = float(margin)
        self.seed = int(seed)
        self.optimizer = optimizer
        self.base_lr = float(base_learning_rate)
        self.decay_steps = learning_rate_decay_examples
        self.decay_rate = learning_rate_decay
        self.exchange, self.grad_sync = exchange, grad_sync
        if grad_sync_mode not in self.GRAD_SYNC_MODES:
            raise ValueError("grad_sync_mode must be one of %s" % (self.GRAD_SYNC_MODES,))
        self._grad_sync_mode = grad_sync_mode
        self.slot0 = int(slot0)
        self.batch_global = self.B if batch_global is None else int(batch_global)
        F = table.feature_size if feature_size is None else feature_size
        if precision in ("auto", None):
            # round 6: the default is the path bench.py times.  An fp16 catalogue is config 4's path; an fp32
            # catalogue takes the split-fp32 products on the bf16 MFMA ("f32x3": fp32 results, 1.7 x the fp32 MFMA's rate)
            # whenever the step's row count allows it (a multiple of 128: the plane kernels' row tiles), else the fp32 MFMA.
            # Say precision="f32" / "f32x3" to pin one.
            if table.data.dtype == torch.float16:
                precision = "bf16"
            else:
                precision = "f32x3" if (self.B * self.rows_per_triplet) % 128 == 0 else "f32"
        if precision not in ("f32", "bf16", "f32x3", "f32x3-3", "f16x2"):
            raise ValueError("precision must be 'auto', 'f32', 'f32x3', 'f16x2' or 'bf16'")
        self.precision = precision
        self.bf16 = precision == "bf16"          # BASELINE config 4: fp16 table + bf16 MFMA
        # fp32 products on the bf16 MFMA: operands as three exact bf16 planes, six plane products (engine_x3;
        # "f32x3-3": the three leading products only -- 16-bit operands, not an fp32 equivalent)
        self.x3 = precision.startswith("f32x3")
        # fp32 products on the fp16 MFMA: operands as two fp16 planes under per-tensor scales, three plane products (engine_f16x2)
        self.h2 = precision == "f16x2"
        if self.bf16 != (table.data.dtype == torch.float16):
            raise ValueError("precision 'bf16' goes with an fp16 FeatureTableF16, 'f32' / 'f32x3' with an fp32 table")
        if self.h2:
            if use_graph and (use_graph == "split" or exchange is not None or grad_sync is not None):
                raise ValueError("precision 'f16x2' replays from a hipGraph on one GPU only (its plane scales are kernel arguments "
                                 "baked into a capture: the check steps run eagerly and a moved scale re-captures, engine_f16x2.py)")
            self._h2_sig = None                          # the scales the captured graphs were recorded with
            self.layout = engine_x3.layout_x3(F, hidden_size, output_size)
            self.params = engine.VNetParams(self.layout, self.device, weight_seed)
            # one GPU: the fused sampler + gather writes the fp16 planes; sharded catalogue: the rows arrive in fp32 through the
            # exchange and the forward pass splits them
            self.ws = engine_f16x2.TowerWorkspaceH2(self.layout, self.R, self.device, planes_in=exchange is None)
        elif self.x3:
            self.layout = engine_x3.layout_x3(F, hidden_size, output_size)
            self.params = engine.VNetParams(self.layout, self.device, weight_seed)
            self.ws = engine_x3.TowerWorkspaceX3(self.layout, self.R, self.device, products=3 if precision.endswith("-3") else 6,
                                                 # the plane buffer IS x_hat on both paths: the fused gather writes planes, and so
                                                 # does the un-permute pass of the row exchange (round 5; a caller's fp32 rows:
                                                 # TowerWorkspaceX3(planes_in=False) splits them in the forward pass)
                                                 planes_in=True,
                                                 # the interleaved copy of x_hat comes from the fused sampler + gather (one GPU)
                                                 kint=None if (exchange is None and not train_table) else False)
            engine_x3.refresh_weights(self.params, self.ws)
        elif self.bf16:
            self.layout = engine_bf16.layout_bf16(F, hidden_size, output_size)
            self.params = engine.VNetParams(self.layout, self.device, weight_seed)
            self.ws = engine_bf16.TowerWorkspaceBF16(self.layout, self.R, self.device)
            engine_bf16.refresh_weights(self.params, self.ws)
        else:
            self.layout = engine.TowerLayout(F, hidden_size, output_size)
            self.params = engine.VNetParams(self.layout, self.device, weight_seed)
            self.ws = engine.TowerWorkspace(self.layout, self.R, self.device)
        dev, i32, f32 = self.device, torch.int32, torch.float32
        self.idx = torch.zeros(self.R, dtype=i32, device=dev)       # [B,3] or [2B] video ids
        self.shift = torch.zeros(1, dtype=i32, device=dev)
        self.pos = torch.zeros(self.B, dtype=f32, device=dev)
        self.neg = torch.zeros(self.B, dtype=f32, device=dev)
        self.hinge = torch.zeros(self.B, dtype=f32, device=dev)
        self.valid = torch.ones(self.B, dtype=torch.uint8, device=dev)
        self.stats = torch.zeros(8, dtype=f32, device=dev)   # loss, mean pos, mean neg, active, variance
        self.oob = torch.zeros(1, dtype=i32, device=dev)     # fused sampler+gather: a pair id outside the table
        self.adam_tickets = ops.new_tickets(dev)             # Adam: last block advances the step counter
        self.var_ws = None                                   # set by enable_variance()
        if mode == "semihard":
            if self.B % 32:
                raise ValueError("semi-hard mining needs a batch that is a multiple of 32")
            # precision f32x3: the score product on the plane kernels with the selection as its epilogue -- no B x 2B
            # score matrix (537 MB at B = 8192) is written or scanned (csrc/gemm_bf16x3.hip; CDML_MINE_FUSED=0: the
            # round-2 form, for A/B runs); the fp32-MFMA and bf16 paths keep the score matrix
            # (precision f16x2: the same epilogue on the fp16 build of the kernel, cdml_semihard_mine_h2)
            self.mine_fused = ((precision == "f32x3" or self.h2) and (2 * self.B) % 256 == 0 and self.layout.Dp % 64 == 0
                               and os.environ.get("CDML_MINE_FUSED", "1") != "0")
            if self.mine_fused:
                Dp = self.layout.Dp
                # (f16x2: the miner's score product on two fp16 planes of the unit rows times 2^14 -- no range to manage)
                self.e3 = (torch.zeros((2 * self.B, 2 * Dp), dtype=torch.float16, device=dev) if self.h2
                           else torch.zeros((2 * self.B, 3 * Dp), dtype=torch.bfloat16, device=dev))
                self.dp = torch.zeros(self.B, dtype=f32, device=dev)
                self.mine_ws = torch.zeros(ops.semihard_mine_x3_workspace(self.B) // 4, dtype=f32, device=dev)
            else:
                self.S = torch.zeros((self.B, 2 * self.B), dtype=f32, device=dev)   # anchor x row dot products
            self.sqn = torch.zeros(2 * self.B, dtype=f32, device=dev)
            self.neg_row = torch.zeros(self.B, dtype=i32, device=dev)
            self.scale = torch.zeros(self.B, dtype=f32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.lr_dev = torch.full((1,), self.base_lr, dtype=f32, device=dev)
        self._lr_host = self.base_lr
        self.global_step = 0
        n = self.layout.numel
        if optimizer == "adam":
            self.m = torch.zeros(n, dtype=f32, device=dev)
            self.v = torch.zeros(n, dtype=f32, device=dev)
        else:
            self.acc = torch.zeros(n, dtype=f32, device=dev)
        self.lars_scratch = torch.zeros(max(ops.lars_scratch_floats(), ops.lars_multi_scratch_floats()), dtype=f32,
                                        device=dev)
        self.clip_gradient_norm = float(clip_gradient_norm)
        self.reg_scale = float(regularization_penalty) * float(l2_penalty)
        self.l2_penalty = float(l2_penalty)
        self.grad_norms = torch.zeros((4, 2), dtype=f32, device=dev)   # per variable: |g|, |w|^2
        self.train_table = bool(train_table)
        if self.train_table:
            if self.bf16 or optimizer != "adam" or precision == "f32x3-3":
                # (config 4's catalogue is fp16: it has no fp32 master rows for Adam to move by 1e-2 * 2^-11 of a value)
                raise ValueError("train_table goes with an fp32 catalogue (precision 'f32x3' or 'f32') and the Adam optimizer")
            if self.x3:
                # the row gradient dLoss/dx_hat = dz1 . W1^T is one more fp32 product on the plane kernels (k-contiguous
                # form: dz1's planes as they are, W1 in its NATURAL orientation [F][hi H | mid H | lo H] -- a third plane
                # copy of W1 that the Adam launch writes with the update, as it does W2's two)
                if getattr(self.ws, "transposed", False) or self.ws.dz1 is None:
                    raise ValueError("train_table needs the row-major activation layout of the f32x3 path")
                self.ws.W1n = torch.zeros((self.layout.Fp, 3 * self.layout.Hp), dtype=torch.bfloat16, device=dev)
                engine_x3.refresh_weights(self.params, self.ws)
            elif self.h2:      # the same sixth product on the fp16 planes: W1 [F][hi H | lo H] at the weights' scale
                self.ws.W1n = torch.zeros((self.layout.Fp, 2 * self.layout.Hp), dtype=torch.float16, device=dev)
            self.tab_m = torch.zeros_like(table.data)
            self.tab_v = torch.zeros_like(table.data)
            self.tab_head = torch.full((table.n_rows,), -1, dtype=i32, device=dev)
            self.tab_next = torch.zeros(self.R, dtype=i32, device=dev)
            self.dxh = torch.zeros((self.R, self.layout.Fp), dtype=f32, device=dev)
            prefetch = False      # a prefetched batch would read rows from before this step's update
            if grad_sync is not None and exchange is None:
                raise ValueError("train_table with grad_sync needs a row-sharded table (exchange): "
                                 "replicated trainable tables would diverge between ranks")
        # several steps' rows per gather launch (the frozen catalogue cannot change in between;
        # a trainable one can, and the sharded path has its own prefetcher)
        kint = self.x3 and getattr(self.ws, "kint", False)
        if gather_ahead in ("auto", None, 0):
            row_bytes = self.ws.x_hat.shape[-1] * self.ws.x_hat.element_size() * (2 if kint else 1)   # what one gathered row writes
            target = 230e6 if self.bf16 else 300e6                                  # (fp16 -> bf16 rows: half the bytes read per byte written)
            gather_ahead = min(4, max(1, int(round(target / (self.R * row_bytes)))))
        self.gather_ahead = max(1, int(gather_ahead))
        if exchange is not None or self.train_table:
            self.gather_ahead = 1
        self._ahead_base = None
        if self.gather_ahead > 1:
            # (fetching the NEXT block on a side stream under this block's GEMMs was measured: the
            # gather left the critical path, but the GEMM it ran beside lost as much -- dropped)
            K = self.gather_ahead
            self._xa = torch.zeros((K,) + tuple(self.ws.x_hat.shape), dtype=self.ws.x_hat.dtype, device=dev)
            self._xka = torch.zeros((K, 3 * self.R * self.layout.Fp), dtype=torch.bfloat16, device=dev) if kint else None
            self._idxa = torch.zeros((K, self.R), dtype=i32, device=dev)
            self._shifta = torch.zeros(K, dtype=i32, device=dev)
            self._select_ahead(0)
        elif kint:
            self._xka = None
            self.ws.xk = torch.zeros(3 * self.R * self.layout.Fp, dtype=torch.bfloat16, device=dev)
        self._graphs = {}
        self._warmed = False
        self._replayed = False                           # the previous step was a graph replay
        # the data-parallel step is enqueue-only too (fixed-capacity exchange, no host counts), so
        # it captures like the single-GPU one: one graph per prefetch buffer
        self.use_graph = "split" if use_graph == "split" else bool(use_graph)
        if self.use_graph and self.train_table and exchange is not None:
            self.use_graph = False                       # scatter_back sizes its scratch on the fly
            logging.getLogger("cdml.train").warning("use_graph ignored: trainable sharded table runs eagerly")
        # (with RCCL inside: under capture the exchange is recorded on the capturing stream, see
        # dist.Prefetcher; verified over RCCL at world size 1, tests/test_gpu_dist.py)
        if self.use_graph and (exchange is not None or grad_sync is not None):
            from . import dist as _cdist
            staged = [h for h in (exchange, grad_sync) if h is not None and h.world > 1
                      and _cdist._host_staged(h.group)]
            if staged:                                   # gloo rehearsal: collectives go through host memory
                self.use_graph = False
                logging.getLogger("cdml.train").warning(
                    "use_graph ignored: host-staged (gloo) collectives cannot be captured; RCCL ones can")
            elif self._needs_capture_groups() and _cdist.capture_groups_supported() is None:
                # the capture-only groups need a default group bound to its device (dist.new_capture_group); a caller
                # that initialised torch.distributed without device_id= (what worked through round 3) steps eagerly
                self.use_graph = False
                logging.getLogger("cdml.train").warning(
                    "use_graph ignored: capturing RCCL collectives needs init_process_group(..., device_id=...); stepping eagerly")
            else:
                self._ensure_capture_groups()
        # row-sharded catalogue: the exchange of step t+1 runs ahead on a side stream
        # into the second x_hat / idx buffer while step t computes
        self.prefetch = None
        self._filled = -1
        if exchange is not None and prefetch:
            from .dist import Prefetcher
            self.prefetch = Prefetcher(self.device)
            self._x = [self.ws.x_hat, torch.zeros_like(self.ws.x_hat)]
            self._idx = [self.idx, torch.zeros_like(self.idx)]
            self._shift = [self.shift, torch.zeros_like(self.shift)]
        if self.use_graph == "split" and self.prefetch is None:
            self.use_graph = True                        # nothing to overlap: one graph per step

    def _needs_capture_groups(self):
        """True when a captured step of this job would issue an RCCL collective (a hook over the nccl backend that is
        not skipped at world size 1)."""
        for h in (self.exchange, self.grad_sync):
            if h is None or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
                continue
            if torch.distributed.get_backend(h.group) != "nccl":
                continue
            if h.world > 1 or not getattr(h, "skip_self", True) or getattr(h, "active", False):
                return True
        return False

    def _ensure_capture_groups(self):
        """Captured collectives go through process groups of their own, which never carry an eager one: the RCCL watchdog's
        work list of a captured communicator is then empty by construction (dist.new_capture_group says why that matters
        -- round 3 slept 0.25 s before a capture instead).  Created at construction when the step is to replay RCCL from
        graphs, and again checked right before any capture (a step switched to graph replay later): a collective call,
        reached by every rank at the same point in the same order."""
        from . import dist as _cdist
        for h in (self.exchange, self.grad_sync):
            if h is None or getattr(h, "capture_group", None) is not None:
                continue
            if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
                continue
            if torch.distributed.get_backend(h.group) != "nccl":
                continue
            if h.world > 1 or not getattr(h, "skip_self", True) or getattr(h, "active", False):
                h.capture_group = _cdist.new_capture_group(h.group)

    @property
    def grad_sync_mode(self):
        return self._grad_sync_mode

    @grad_sync_mode.setter
    def grad_sync_mode(self, mode):
        if mode not in self.GRAD_SYNC_MODES:
            raise ValueError("grad_sync_mode must be one of %s" % (self.GRAD_SYNC_MODES,))
        if mode != self._grad_sync_mode:
            self._grad_sync_mode = mode
            self._graphs = {}                            # captured steps recorded the other form

    # ---------------------------------------------------------------- pieces --
    def _fill(self, b, step):
        """Sample step `step` (= the current step + 1) and exchange its rows into buffer b.
        Eager: the step is an immediate (the side stream runs ahead of the device counter).
        Inside a graph capture the kernels read the device counter + 1 instead, so replays follow
        it; the branch completes before the all-reduce is issued, hence before the optimizer
        advances the counter."""
        sd = None
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            step, sd = 1, self.step_dev
        if _MODES[self.mode] == MODE_UNIFORM:
            ops.sample_uniform(self.pairs, self.table.n_rows_global, self.seed, step, self.B,
                               self._idx[b], slot0=self.slot0, batch_global=self.batch_global, step_dev=sd)
        else:
            ops.sample_inbatch(self.pairs, self.seed, step, self.B, self._idx[b], self._shift[b],
                               slot0=self.slot0, batch_global=self.batch_global, step_dev=sd)
        self.exchange.gather(self.table, self._idx[b], self._x[b])

    def _ahead_offset(self):
        """Position of the current step inside the block of steps the last gather launch
        fetched; 0 = this step launches the gather."""
        t, b = self.global_step, self._ahead_base
        if self.gather_ahead == 1 or b is None or not (b <= t < b + self.gather_ahead):
            return 0
        return t - b

    def _select_ahead(self, off):
        self.ws.x_hat, self.idx, self.shift = self._xa[off], self._idxa[off], self._shifta[off:off + 1]
        if getattr(self, "_xka", None) is not None:
            self.ws.xk = self._xka[off]

    def _gather_block(self, step=None):
        """Sample + gather the steps step .. step+gather_ahead-1 in one launch (step None: the
        device counter's value when the kernel runs)."""
        ops.sample_gather(_MODES[self.mode], self.pairs, self.seed, step, self.B, self.table.data,
                          self.table.feature_size, self._idxa, self._xa, shift_out=self._shifta,
                          slot0=self.slot0, batch_global=self.batch_global,
                          step_dev=self.step_dev if step is None else None, n_steps=self.gather_ahead,
                          oob_flag=self.oob, x_ki=getattr(self, "_xka", None))

    def fetch(self):
        """Sampler + gather (+ input l2norm): fills ws.x_hat and self.idx."""
        m = _MODES[self.mode]
        if self.gather_ahead > 1:
            off = self._ahead_offset()
            if off == 0:
                self._gather_block()
                self._ahead_base = self.global_step
            self._select_ahead(off)
            return
        if self.prefetch is not None:
            t, b = self.global_step, self.global_step % 2
            if t == 0 or self._filled != t:
                self.prefetch.launch(b, lambda: self._fill(b, t))       # cold start / after a resume
            self.prefetch.acquire(b)
            self.ws.x_hat, self.idx, self.shift = self._x[b], self._idx[b], self._shift[b]
            return
        if self.exchange is None:
            ops.sample_gather(m, self.pairs, self.seed, None, self.B, self.table.data,
                              self.table.feature_size, self.idx, self.ws.x_hat,
                              shift_out=self.shift, slot0=self.slot0,
                              batch_global=self.batch_global, step_dev=self.step_dev, oob_flag=self.oob,
                              x_ki=self.ws.xk if (self.x3 and getattr(self.ws, "kint", False)) else None)
        else:
            if m == MODE_UNIFORM:
                ops.sample_uniform(self.pairs, self.table.n_rows_global, self.seed, None, self.B,
                                   self.idx, slot0=self.slot0, batch_global=self.batch_global,
                                   step_dev=self.step_dev)
            else:
                ops.sample_inbatch(self.pairs, self.seed, None, self.B, self.idx, self.shift,
                                   slot0=self.slot0, batch_global=self.batch_global,
                                   step_dev=self.step_dev)
            self.exchange.gather(self.table, self.idx, self.ws.x_hat)

    def enable_variance(self, on=True):
        """Also compute build_graph's ``variance`` summary (calc_var, train.py:67-71,151) each
        step: stats[4]."""
        L = self.layout
        self.var_ws = (torch.zeros(ops.vnet_tail_workspace_floats(self.B, L.Dp), dtype=torch.float32,
                                   device=self.device) if on else None)
        self._graphs = {}

    def variance(self):
        if self.var_ws is None:
            raise RuntimeError("call enable_variance() before the step")
        # (the tail kernel averages over the PADDED embedding width it is given, L.Dp; the reference's calc_var over the D real
        # columns -- the padding columns are zero, so the two differ by exactly Dp / D where a layout pads D: the plane layouts
        # at D < 256.  Found by running build_graph's switches on the plane paths, round 6.)
        return float(self.stats[4].item()) * self.layout.Dp / self.layout.D

    def forward_loss(self, with_grad=True):
        # precision f16x2: on its check steps (0, 1, 2, 4 .. 64, then every 64th) the plane scales are re-derived -- the weights'
        # and the hidden layer's before the forward pass, the gradients' once the loss tail has written dz2 (two small
        # device-to-host copies; engine_f16x2.PlaneScales)
        due = self.h2 and self.ws.scales.due(self.global_step)
        if due:
            engine_f16x2.observe_weights(self.params, self.ws)
        self._forward_loss(with_grad)
        if due and with_grad:
            engine_f16x2.observe_gradients(self.params, self.ws)

    def _forward_loss(self, with_grad=True):
        # uniform / in-batch negatives: one launch normalises z, takes the loss and starts the
        # backward pass (semi-hard mining needs every embedded row first: separate kernels)
        fused = with_grad and self.mode != "semihard"
        # (semi-hard mining on the plane kernels: the miner's prep launch normalises z itself -- round 6)
        mine_norm = (self.mode == "semihard" and getattr(self, "mine_fused", False)
                     and os.environ.get("CDML_MINE_NORM", "1") != "0")
        if self.h2:
            engine_f16x2.tower_forward(self.params, self.ws, normalize=not (fused or mine_norm))
        elif self.x3:
            engine_x3.tower_forward(self.params, self.ws, normalize=not (fused or mine_norm))
        elif self.bf16:
            engine_bf16.tower_forward(self.params, self.ws, normalize=not fused)
        else:
            engine.tower_forward(self.params, self.ws, normalize=not fused)
        L = self.layout
        if fused:
            ops.vnet_tail(0 if self.mode == "uniform" else 1, self.ws.z, self.idx, self.shift, self.B, L.Dp,
                          self.margin, self.ws.e, self.pos, self.neg, self.hinge, self.ws.dz2, valid=self.valid,
                          stats=self.stats, var_ws=self.var_ws,
                          dz2_bf16=self.ws.dz2_bf if self.bf16 else self.ws.dz2_3 if self.x3 else self.ws.dz2_2 if self.h2 else None,
                          plane_bf=L.Dp if (self.x3 or self.h2) else 0, h2_scale=self.ws.scales.dz2 if self.h2 else 0.0)
            self.ws.tail_done = True
            self.ws.dz2_planes_done = self.x3 or self.h2
            return
        de = self.ws.de if with_grad else None
        if self.mode == "uniform":
            ops.triplet_hinge(self.ws.e, self.B, L.Dp, self.margin, self.pos, self.neg, self.hinge,
                              self.stats, de)
        elif self.mode == "semihard":
            e = self.ws.e
            if self.mine_fused:
                ops.semihard_mine_x3(e, self.idx, self.B, L.Dp, self.e3, L.Dp, self.sqn, self.dp, self.mine_ws, self.neg_row,
                                     z=self.ws.z if mine_norm else None, h2_scale=engine_f16x2.X_SCALE if self.h2 else 0.0)
            else:
                # S[i][c] = <anchor_i, row_c>: the data-gradient GEMM (x @ W^T) with no mask
                ops.fc_bwd_data(e[0::2], e, None, self.S, self.B, 2 * self.B, L.Dp)
                ops.semihard_select(self.S, e, self.idx, self.B, L.Dp, self.sqn, self.neg_row)
            if with_grad and os.environ.get("CDML_INDEXED_TAIL", "1") != "0":
                # round 6: the rest of the tail rides in the gradient launch -- every finished row gradient goes through the
                # l2norm backward (+ leaky-relu') into dz2 and its operand copy (bf16 / three planes) on the spot: two
                # launches fewer per step, bit-identical (CDML_INDEXED_TAIL=0: the separate launches, for A/B runs)
                ops.triplet_hinge_indexed(e, self.neg_row, self.B, L.Dp, self.margin, self.pos, self.neg, self.hinge,
                                          self.scale, self.stats, de, z=self.ws.z, dz2=self.ws.dz2,
                                          dz2_bf16=self.ws.dz2_bf if self.bf16 else self.ws.dz2_3 if self.x3 else None,
                                          plane_bf=L.Dp if self.x3 else 0)
                self.ws.tail_done = True
                self.ws.dz2_planes_done = self.x3
            else:
                ops.triplet_hinge_indexed(e, self.neg_row, self.B, L.Dp, self.margin, self.pos, self.neg,
                                          self.hinge, self.scale, self.stats, de)
        else:
            ops.triplet_hinge_inbatch(self.ws.e, self.idx, self.shift, self.B, L.Dp, self.margin,
                                      self.pos, self.neg, self.hinge, self.valid, self.stats, de)

    def backward(self, after_w1=None):
        if self.h2:
            engine_f16x2.tower_backward(self.params, self.ws, after_w1=after_w1)
        elif self.x3:
            engine_x3.tower_backward(self.params, self.ws, after_w1=after_w1)
        elif self.bf16:
            engine_bf16.tower_backward(self.params, self.ws, after_w1=after_w1)
        else:
            engine.tower_backward(self.params, self.ws, after_w1=after_w1)

    def update_table(self):
        """dLoss/d x_hat = dz1 . W1^T for the gathered rows, then the lazy-Adam update of the
        catalogue rows they came from (on their owners when the table is sharded).  Runs
        before the dense update: it needs this step's W1 and step counter."""
        L, p, t = self.layout, self.params, self.table
        if self.h2:
            sc = self.ws.scales
            ops.gemm_f16x2_nt(ops.BE_F32, self.ws.dz1, L.Hp, self.ws.W1n, L.Hp, self.dxh, self.R, L.Fp, L.Hp,
                              1.0 / (sc.dz1 * sc.w1))
        elif self.x3:
            ops.gemm_bf16x3_nt(ops.BE_F32, self.ws.dz1, L.Hp, self.ws.W1n, L.Hp, self.dxh, self.R, L.Fp, L.Hp,
                               products=self.ws.products)
        else:
            ops.fc_bwd_data(self.ws.dz1, p.W1, None, self.dxh, self.R, L.Fp, L.Hp)
        idx, rows = self.idx, self.dxh
        if self.exchange is not None:
            idx, rows = self.exchange.scatter_back(self.dxh)
            if idx.numel() == 0:
                return
            if self.tab_next.numel() < idx.numel():
                self.tab_next = torch.zeros(idx.numel(), dtype=torch.int32, device=self.device)
        # dxh carries the mean over THIS rank's batch; the dense gradients are averaged over
        # ranks, so the row gradients (summed on the owner) take the same 1/world
        world = self.grad_sync.world if self.grad_sync is not None else 1
        ops.table_adam_rows(t.data, t.row0, t.feature_size, idx, rows, self.tab_m, self.tab_v,
                            self.tab_head, self.tab_next, 0.0, 1, lr_dev=self.lr_dev, t_dev=self.step_dev,
                            grad_scale=1.0 / world)

    def reg_loss(self):
        """build_graph's reg_loss summary (train.py:133-136): sum over the weight matrices of
        l2_penalty*|W|^2/2, as of the last step (needs regularization_penalty or clipping on)."""
        n = self.grad_norms.cpu()
        return float(self.l2_penalty * (n[0, 1] + n[2, 1]) / 2.0)

    def summaries(self):
        """The scalars build_graph and HingeLoss hand to TensorBoard (train.py:154-160: loss, reg_loss, variance,
        final_learning_rate; losses.py:40-41: mean_pos_dist, mean_neg_dist), as of the last step, under the reference's
        names -- ONE device-to-host copy (the step's own stats words and the weight matrices' row norms; the step
        itself is untouched: `variance` needs enable_variance() before the step and is None otherwise).  Synchronises:
        the trainer calls it at its evaluation cadence, where it waits for the device anyway."""
        L, p = self.layout, self.params
        if getattr(self, "_wsq", None) is None:
            self._wsq = torch.zeros(8 + L.Fp + L.Hp, dtype=torch.float32, device=self.device)
        # reg_loss = sum over the weight matrices of l2_penalty * |W|^2 / 2 (models.py:28, train.py:133-136): the rows'
        # squared norms by the HIP kernel, added up on the host
        ops.row_sqnorm(p.W1, L.Hp, self._wsq[8:8 + L.Fp])
        ops.row_sqnorm(p.W2, L.Dp, self._wsq[8 + L.Fp:])
        self._wsq[:8].copy_(self.stats)
        h = self._wsq.cpu().double()
        return {"loss": float(h[0]), "reg_loss": float(self.l2_penalty * h[8:].sum() / 2.0),
                "variance": float(h[4]) * L.Dp / L.D if self.var_ws is not None else None,
                "final_learning_rate": float(self._lr_host), "mean_pos_dist": float(h[1]), "mean_neg_dist": float(h[2]),
                "active_triplets": float(h[3])}

    def apply_gradients(self):
        p = self.params
        if self.clip_gradient_norm > 0.0 or self.reg_scale != 0.0:
            for i, (off, n) in enumerate(p.segments()):    # W1, b1, W2, b2: weights carry the regulariser
                ops.grad_prepare(p.grad[off:off + n], p.flat[off:off + n], self.reg_scale if i % 2 == 0 else 0.0,
                                 self.clip_gradient_norm, self.lars_scratch, self.grad_norms[i])
        if self.optimizer == "adam" and self.h2:
            # two fp16 planes: the same two launches, the copies written as the planes of W * its scale (the scales as of
            # this step: engine_f16x2.observe_weights re-splits the weights itself when it moves one)
            L, o, ws = self.layout, self.layout.offsets, self.ws
            mat = lambda t, i, r, c: t[o[i]:o[i] + r * c].view(r, c)
            kw = dict(lr_dev=self.lr_dev, t_dev=self.step_dev)
            b1, b2 = slice(o[1], o[1] + L.Hp), slice(o[3], o[3] + L.Dp)
            vec = lambda sl: (p.flat[sl], p.grad[sl], self.m[sl], self.v[sl])
            w1n = getattr(ws, "W1n", None)                 # trainable table: W1's planes in their natural orientation too
            ops.adam_matrix_bf16(p.W1, mat(p.grad, 0, L.Fp, L.Hp), mat(self.m, 0, L.Fp, L.Hp),
                                 mat(self.v, 0, L.Fp, L.Hp), 0.0, 1, wt=ws.W1T, plane_t=L.Fp, bias=vec(b1),
                                 h2_scale=ws.scales.w1, **(dict(kw, wc=w1n, plane_c=L.Hp) if w1n is not None else kw))
            ops.adam_matrix_bf16(p.W2, mat(p.grad, 2, L.Hp, L.Dp), mat(self.m, 2, L.Hp, L.Dp),
                                 mat(self.v, 2, L.Hp, L.Dp), 0.0, 1, wt=ws.W2T, plane_t=L.Hp, wc=ws.W2, plane_c=L.Dp,
                                 bias=vec(b2), advance_tickets=self.adam_tickets, h2_scale=ws.scales.w2, **kw)
        elif self.optimizer == "adam" and self.x3:
            # split-fp32 precision: as on the config-4 path two launches, each weight matrix with its bias vector;
            # the update writes the plane copies the GEMMs read (W1^T; W2^T and W2)
            L, o, ws = self.layout, self.layout.offsets, self.ws
            mat = lambda t, i, r, c: t[o[i]:o[i] + r * c].view(r, c)
            kw = dict(lr_dev=self.lr_dev, t_dev=self.step_dev)
            b1, b2 = slice(o[1], o[1] + L.Hp), slice(o[3], o[3] + L.Dp)
            vec = lambda sl: (p.flat[sl], p.grad[sl], self.m[sl], self.v[sl])
            w1n = getattr(ws, "W1n", None)                 # trainable table: W1's planes in their natural orientation too
            ops.adam_matrix_bf16(p.W1, mat(p.grad, 0, L.Fp, L.Hp), mat(self.m, 0, L.Fp, L.Hp),
                                 mat(self.v, 0, L.Fp, L.Hp), 0.0, 1, wt=ws.W1T, plane_t=L.Fp, bias=vec(b1),
                                 **(dict(kw, wc=w1n, plane_c=L.Hp) if w1n is not None else kw))
            ops.adam_matrix_bf16(p.W2, mat(p.grad, 2, L.Hp, L.Dp), mat(self.m, 2, L.Hp, L.Dp),
                                 mat(self.v, 2, L.Hp, L.Dp), 0.0, 1, wt=ws.W2T, plane_t=L.Hp, wc=ws.W2, plane_c=L.Dp,
                                 bias=vec(b2), advance_tickets=self.adam_tickets, **kw)
        elif self.optimizer == "adam" and self.bf16:
            # config-4 precision: two launches -- each weight matrix with its bias vector; the update
            # also writes the bf16 operand copies the GEMMs read (no separate transposes / cast), and
            # the last block of the second launch advances the step counter
            L, o, ws = self.layout, self.layout.offsets, self.ws
            mat = lambda t, i, r, c: t[o[i]:o[i] + r * c].view(r, c)
            kw = dict(lr_dev=self.lr_dev, t_dev=self.step_dev)
            b1, b2 = slice(o[1], o[1] + L.Hp), slice(o[3], o[3] + L.Dp)
            vec = lambda sl: (p.flat[sl], p.grad[sl], self.m[sl], self.v[sl])
            ops.adam_matrix_bf16(p.W1, mat(p.grad, 0, L.Fp, L.Hp), mat(self.m, 0, L.Fp, L.Hp),
                                 mat(self.v, 0, L.Fp, L.Hp), 0.0, 1, wt=ws.W1T, bias=vec(b1), **kw)
            ops.adam_matrix_bf16(p.W2, mat(p.grad, 2, L.Hp, L.Dp), mat(self.m, 2, L.Hp, L.Dp),
                                 mat(self.v, 2, L.Hp, L.Dp), 0.0, 1, wt=ws.W2T, wc=ws.W2, bias=vec(b2),
                                 advance_tickets=self.adam_tickets, **kw)
        elif self.optimizer == "adam":
            # the step counter advances inside the same launch
            ops.adam_step(p.flat, p.grad, self.m, self.v, 0.0, 1, lr_dev=self.lr_dev,
                          t_dev=self.step_dev, advance_tickets=self.adam_tickets)
        elif self.x3 or self.bf16 or self.h2:
            # LARS / momentum on the plane (f32x3, f16x2) or bf16 (config 4) paths: two matrix launches, each weight matrix with
            # its bias vector, writing the operand copies the GEMMs read with the update (as the Adam launches above do --
            # round 3 ran separate split / transpose launches after the optimizer); the last one advances the step counter
            L, o, ws = self.layout, self.layout.offsets, self.ws
            pt1, pt2, pc2 = (L.Fp, L.Hp, L.Dp) if (self.x3 or self.h2) else (0, 0, 0)
            s1, s2 = (ws.scales.w1, ws.scales.w2) if self.h2 else (0.0, 0.0)     # (f16x2: fp16 planes of W * its scale)
            if self.optimizer == "lars":
                segs = p.segments()
                ops.lars_multi_norms(p.flat, p.grad, segs, self.lars_scratch)
                kw = dict(lr_dev=self.lr_dev)
                ops.lars_matrix(p.flat, p.grad, self.acc, segs, 0, 1, L.Fp, L.Hp, 0.0, self.lars_scratch, wt=ws.W1T,
                                plane_t=pt1, h2_scale=s1, **kw)
                ops.lars_matrix(p.flat, p.grad, self.acc, segs, 2, 3, L.Hp, L.Dp, 0.0, self.lars_scratch, wt=ws.W2T, wc=ws.W2,
                                plane_t=pt2, plane_c=pc2, step_dev=self.step_dev, tickets=self.adam_tickets, h2_scale=s2, **kw)
            else:
                mat = lambda t, i, r, c: t[o[i]:o[i] + r * c].view(r, c)
                b1, b2 = slice(o[1], o[1] + L.Hp), slice(o[3], o[3] + L.Dp)
                vec = lambda sl: (p.flat[sl], p.grad[sl], self.acc[sl])
                ops.momentum_matrix(p.W1, mat(p.grad, 0, L.Fp, L.Hp), mat(self.acc, 0, L.Fp, L.Hp), 0.0, wt=ws.W1T,
                                    plane_t=pt1, lr_dev=self.lr_dev, bias=vec(b1), h2_scale=s1)
                ops.momentum_matrix(p.W2, mat(p.grad, 2, L.Hp, L.Dp), mat(self.acc, 2, L.Hp, L.Dp), 0.0, wt=ws.W2T, wc=ws.W2,
                                    plane_t=pt2, plane_c=pc2, lr_dev=self.lr_dev, bias=vec(b2), step_dev=self.step_dev,
                                    tickets=self.adam_tickets, h2_scale=s2)
        elif self.optimizer == "momentum":
            ops.momentum_step(p.flat, p.grad, self.acc, 0.0, 0.9, True, lr_dev=self.lr_dev)
            ops.step_advance(self.step_dev)
        else:
            # LARS: one trust ratio per variable, all four variables in two launches; the second
            # also advances the step counter
            ops.lars_multi(p.flat, p.grad, self.acc, p.segments(), 0.0, self.lars_scratch, lr_dev=self.lr_dev,
                           step_dev=self.step_dev, tickets=self.adam_tickets)

    def _enqueue(self):
        self.fetch()
        if self.prefetch is not None:
            # next step's rows right away (the sampler is counter-based): the exchange runs under
            # this step's forward GEMMs, and the gradient all-reduce below is issued only after
            # it has finished, so the two communicators never run side by side and every rank
            # executes its collectives in the same order
            t, b = self.global_step, self.global_step % 2
            self.prefetch.launch(1 - b, lambda: self._fill(1 - b, t + 1))
            self._filled = t + 1
        self.forward_loss()
        self._backward_and_update(b if self.prefetch is not None else None)

    def _backward_and_update(self, b):
        """Backward (with the gradient all-reduce hooks), then the optimizer.  ``b``: the prefetch
        buffer this step computes on (None without a prefetcher)."""
        if self.grad_sync is None:
            self.backward()
        elif self._grad_sync_mode == "single":
            # the single-GPU backward (both weight gradients in one stream-K launch), then one
            # all-reduce of the whole flat gradient, issued once the exchange of step t+1 is done
            self.backward()
            if b is not None:
                self.prefetch.wait_ready(1 - b)
            self.grad_sync.finish([self.grad_sync.start(self.params.grad, 0, self.layout.numel)])
        elif self._grad_sync_mode == "two":
            if b is not None:
                self.prefetch.wait_ready(1 - b)
            n1 = self.layout.offsets[2]
            handles = []
            (engine_f16x2 if self.h2 else engine_x3 if self.x3 else engine_bf16 if self.bf16 else engine).tower_backward(
                self.params, self.ws, after_w1=lambda: handles.append(self.grad_sync.start(self.params.grad, 0, n1)))
            handles.append(self.grad_sync.start(self.params.grad, n1, self.layout.numel))
            self.grad_sync.finish(handles)
        else:
            if b is not None:
                self.prefetch.wait_ready(1 - b)
            # buckets: [dW1|db1] (85 % of the bytes) is all-reduced while later GEMMs run, in two
            # row blocks -- the first under the second block's GEMM, the second under the dW2
            # GEMM -- and [dW2|db2] right after its GEMM; the optimizer waits for all of them
            n1 = self.layout.offsets[2]
            handles = []
            (engine_f16x2 if self.h2 else engine_x3 if self.x3 else engine_bf16 if self.bf16 else engine).tower_backward(
                self.params, self.ws, w1_chunks=2,
                after_w1_chunk=lambda lo, hi: handles.append(self.grad_sync.start(self.params.grad, lo, hi)))
            handles.append(self.grad_sync.start(self.params.grad, n1, self.layout.numel))
            self.grad_sync.finish(handles)
        if b is not None:
            self.prefetch.release(b)                 # backward was the last reader of x_hat[b]
        if self.train_table:
            self.update_table()
        self.apply_gradients()

    # ------------------------------------------------------------------ step --
    def step(self):
        """Enqueue one training step (no host sync).  Loss etc. land in
        ``self.stats`` (device)."""
        lr = exponential_decay(self.base_lr, self.global_step, self.decay_steps, self.decay_rate)
        if lr != self._lr_host:                         # staircase: rare
            self.lr_dev.fill_(lr)
            self._lr_host = lr
        # precision f16x2 under use_graph: a check step of the plane scales (host reads) runs eagerly, and graphs recorded with
        # scales that have moved since are dropped (re-captured on the next replay step)
        h2_eager = self.h2 and bool(self.use_graph) and self.ws.scales.due(self.global_step)
        if self.use_graph == "split" and self._warmed and self.prefetch is not None:
            self._step_split(self.global_step)
            self._replayed = True
        elif self.use_graph and self._warmed and not h2_eager:   # the first step of a process runs eagerly
            t = self.global_step                         # (it loads the kernels), also after a resume
            if self.prefetch is not None:
                # one graph per prefetch buffer; the rows of step t were fetched by step t-1 (an
                # eager step handed them over through an event, a replay through stream order)
                key = t % 2
                if self._filled != t:
                    raise RuntimeError("graph replay needs the previous step's prefetch (step %d)" % t)
                self.prefetch.acquire(key)
            else:
                key = self._ahead_offset()               # one graph per position in the gather block
            if key not in self._graphs:
                self._graphs[key] = self._capture()      # (capturing records, it does not run)
            if self.prefetch is not None:
                self.ws.x_hat, self.idx, self.shift = self._x[key], self._idx[key], self._shift[key]
                self._filled = t + 1
            elif self.gather_ahead > 1:
                if key == 0:
                    self._ahead_base = t
                self._select_ahead(key)
            self._graphs[key].replay()
            self._replayed = True
        else:
            if self.prefetch is not None and self._replayed:
                # replay -> eager: a replay records no buffer-free events, so order the side stream
                # behind the replays still queued on the compute stream (and the reverse)
                self.prefetch.drain()
            self._replayed = False
            self._enqueue()
            self._warmed = True
            if self.h2 and self.use_graph:
                sig = tuple(sorted(self.ws.scales.state().items()))
                if sig != self._h2_sig:
                    self._graphs, self._h2_sig = {}, sig
        self.global_step += 1

    def _capture(self, fn=None, origin=None):
        """Record ``fn`` (default: the whole step) into a hipGraph.  ``origin``: the stream the capture
        starts on (default: a fresh one) -- RCCL's communicator stream is then one fork from it."""
        if origin is not None:
            # the invariant the round-5 abort taught: no stream that ever carried an eager RCCL collective enters a capture
            # (the current stream is only waited on before / after the capture, it is not part of it; a fresh origin cannot
            # have carried anything).  Checked before anything else touches the device.
            from .dist import EagerCollectiveStreams
            EagerCollectiveStreams.assert_clean_origin(origin)
        torch.cuda.synchronize(self.device)
        self._ensure_capture_groups()
        # (RCCL collectives recorded below go through the hooks' capture-only process groups -- dist.new_capture_group:
        # no eager collective is ever issued on them, so the watchdog thread has nothing of theirs to poll while their
        # communicator streams are inside this capture; the eager steps' groups are never captured)
        side = torch.cuda.Stream(self.device) if origin is None else origin
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: the RCCL watchdog thread of torch.distributed may touch
            # the runtime while this thread captures
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                (fn or self._enqueue)()
        torch.cuda.current_stream(self.device).wait_stream(side)
        return g

    def _step_split(self, t):
        """One data-parallel step as three graph replays (``use_graph="split"``): the exchange of step
        t+1 on the prefetch stream, forward + loss, then -- once that exchange has finished --
        backward + all-reduce + optimizer on the compute stream; the events between them are recorded
        eagerly, exactly where the eager step records them.  The exchange graph reads the device step
        counter (+1): it runs after the optimizer of step t-1 advanced it (the buffer-free event) and
        before the optimizer of step t does (which waits for the exchange)."""
        pf, b = self.prefetch, t % 2
        if self._filled != t:
            raise RuntimeError("graph replay needs the previous step's prefetch (step %d)" % t)
        G = self._graphs
        if ("E", 1 - b) not in G:
            # Capture origin: a stream of its OWN, never the prefetch stream (rounds 3-4).  torch runs a synchronous eager
            # collective on the CURRENT stream and records its work's end event there, so the eager exchange of the first
            # step left such an event on pf.stream; a capture that starts on pf.stream within the RCCL watchdog's poll
            # period (100 ms) then makes the watchdog query an event "last recorded in a capturing stream" and the process
            # ends (round 5: test_data_parallel_step_replays_from_hipgraph_over_rccl failed once in four suite runs).  The
            # graph is still REPLAYED on pf.stream; RCCL's communicator stream is one fork from the origin either way.
            if getattr(self, "_ex_origin", None) is None:
                self._ex_origin = torch.cuda.Stream(self.device)
            G[("E", 1 - b)] = self._capture(lambda: self._fill(1 - b, None), origin=self._ex_origin)
        self.ws.x_hat, self.idx, self.shift = self._x[b], self._idx[b], self._shift[b]
        if ("F", b) not in G:
            G[("F", b)] = self._capture(self.forward_loss)
            G[("B", b)] = self._capture(lambda: self._backward_and_update(None))
        pf.acquire(b)                                    # rows of step t (fetched during step t-1)
        with torch.cuda.stream(pf.stream):
            if pf._released[1 - b]:
                pf.stream.wait_event(pf.free[1 - b])     # step t-1 is done with that buffer (and has advanced the counter)
            G[("E", 1 - b)].replay()
            pf.ready[1 - b].record(pf.stream)
        self._filled = t + 1
        G[("F", b)].replay()
        pf.wait_ready(1 - b)                             # the all-reduce is issued only after the exchange
        G[("B", b)].replay()
        pf.release(b)

    def check_inputs(self, extra=None):
        """Host check of the device-side input flags (synchronises; called by ``loss()``, by the
        trainer before it saves or evaluates, by bench.py after the timed region): a pair id
        outside the catalogue (the reference's IndexError, inputs.py:158) and, on a row-sharded
        catalogue, an exchange segment that overflowed (the step then trained on NaN rows).
        Data-parallel runs: a COLLECTIVE call -- the flags are reduced (maximum) over the gradient
        all-reduce's group first, because one rank's overflow reaches every rank's weights through
        the gradient average: every rank raises, at the same point, whichever rank overflowed.  So
        every rank has to call it at the same point of its program, whatever its own arguments are
        (``Trainer.save`` calls it before it looks at its rank-local ``checkpoint_dir``).
        ``extra`` (int32[1] device flag or None): one more per-rank flag reduced by the same
        collective; its maximum over the ranks is returned (None -> 0)."""
        ex = self.exchange
        if self.grad_sync is not None and self.grad_sync.world > 1:
            from .dist import reduce_input_flags
            r = reduce_input_flags(self.oob, None if ex is None else ex.overflow, self.grad_sync.group,
                                   self.grad_sync.world, extra=extra)
            oob, over, bad = r[:3]
            extra_max = r[3] if extra is not None else 0
            where = " (on at least one rank)"
        else:
            oob = int(self.oob.item())
            f = 0 if (ex is None or ex.overflow is None) else int(ex.overflow.item())
            over, bad, where = f & 1, (f >> 1) & 1, ""
            extra_max = int(extra.item()) if extra is not None else 0
        if oob or bad:
            raise IndexError("a co-watch pair id lies outside the %d-row catalogue%s" % (self.table.n_rows_global, where))
        if over:
            raise RuntimeError("row exchange: a peer segment overflowed%s (requests are skewed towards one shard): the step "
                               "trained on NaN rows; raise RowExchange(capacity_factor=%.2f)" % (where, ex.capacity_factor))
        return extra_max

    def weights_nonfinite_flag(self):
        """int32[1] on the device: 1 when a weight (or, with a trainable table, a row of this rank's shard) is not
        finite -- what a checkpoint would WRITE, as opposed to the loss of the batch before the update."""
        bad = ~torch.isfinite(self.params.flat).all()
        if self.train_table:
            bad = bad | ~torch.isfinite(self.table.data).all()
        return bad.to(torch.int32).view(1)

    def loss(self, check=True):
        """Host value of the last step's mean hinge loss (synchronises; also reads the input
        flags, so an exchange overflow raises here instead of returning NaN).
        Data-parallel runs (``grad_sync.world > 1``): with ``check`` this is a COLLECTIVE call
        (``check_inputs`` reduces the flags over the ranks) -- every rank must make it at the same
        point; ``loss(check=False)`` is the rank-local read for logging on one rank only."""
        v = float(self.stats[0].item())
        if check:
            self.check_inputs()
        return v

    # ------------------------------------------------------------ checkpoint --
    def state_dict(self):
        """Weights under their slim variable names (fully_connected{,_1}/{weights,
        biases}), optimizer slots, step counter and sampler state."""
        L = self.layout
        slots = {"m": self.m, "v": self.v} if self.optimizer == "adam" else {"acc": self.acc}
        state = {"layout": (L.F, L.H, L.D), "variables": self.params.state_dict(),
                 "optimizer": self.optimizer, "slots": {k: t.detach().cpu().clone() for k, t in slots.items()},
                 "global_step": self.global_step, "seed": self.seed, "mode": self.mode,
                 "batch_size": self.B, "margin": self.margin}
        if self.train_table:                             # this rank's shard and its Adam states
            state["table"] = {"row0": self.table.row0, "rows": self.table.data.detach().cpu().clone(),
                              "m": self.tab_m.detach().cpu().clone(), "v": self.tab_v.detach().cpu().clone()}
        if self.h2:                                      # precision f16x2: the plane scales are state (a resumed run keeps the
            sc = self.ws.scales                          # straight run's bits only with the scales that run would hold)
            state["plane_scales"] = dict(sc.state(), calibrated=sc.calibrated, changes=sc.changes, last=dict(sc.last))
        return state

    def load_state_dict(self, state):
        if tuple(state["layout"]) != (self.layout.F, self.layout.H, self.layout.D):
            raise ValueError("checkpoint layout %s does not match the model" % (state["layout"],))
        if state["optimizer"] != self.optimizer:
            raise ValueError("checkpoint optimizer is %s" % state["optimizer"])
        self.params.load(*[state["variables"][n] for n in engine.VNetParams.NAMES])
        for k, t in state["slots"].items():
            getattr(self, k).copy_(t.to(self.device))
        if self.train_table and "table" in state:
            if int(state["table"]["row0"]) != self.table.row0 or state["table"]["rows"].shape != self.table.data.shape:
                raise ValueError("checkpointed table shard does not match this rank's shard")
            self.table.data.copy_(state["table"]["rows"].to(self.device))
            self.tab_m.copy_(state["table"]["m"].to(self.device))
            self.tab_v.copy_(state["table"]["v"].to(self.device))
        if self.bf16:                                    # the GEMMs read the bf16 copies, not the masters
            engine_bf16.refresh_weights(self.params, self.ws)
        if self.h2:
            sc, saved = self.ws.scales, state.get("plane_scales")
            if saved:                                    # the checkpointed run's scales, and the weights' planes at them
                for k in ("w1", "w2", "h1", "dz2", "dz1"):
                    setattr(sc, k, float(saved[k]))
                sc.calibrated, sc.changes, sc.last = bool(saved["calibrated"]), int(saved["changes"]), dict(saved["last"])
                engine_f16x2.refresh_weights(self.params, self.ws)
            else:
                sc.calibrated = False                    # (a checkpoint of another precision: the next step calibrates)
        if self.x3:
            engine_x3.refresh_weights(self.params, self.ws)
        self.global_step = int(state["global_step"])
        self.step_dev.fill_(self.global_step)
        self.seed = int(state["seed"])
        self._graphs = {}
        self._ahead_base = None
        self._filled = -1
        # the first step after a resume runs eagerly (and refills the prefetch buffer): the side
        # stream may still be filling a buffer for the old step
        self._warmed = False
        self._replayed = False
        if self.prefetch is not None:
            self.prefetch.drain()


class Trainer:
    """The reference's training loop and model-selection policy
    (train.py:177-336) over ``TrainStep``:

      * runs until the pair stream is exhausted (``num_epochs`` passes, final
        partial batch dropped) -- train.py:300-306;
      * every ``eval_step`` steps embeds the held-out rows (``Prediction``) and
        computes ``Evaluation.mean_dist`` -- train.py:224-231;
      * before ``check_stop_step`` evaluations only reset the patience counter;
        afterwards an improvement saves a checkpoint (best-only, one kept) --
        train.py:232-241, 275;
      * early stop when more than ``require_improve_num`` evaluations passed
        without improvement past ``check_stop_step`` -- train.py:307-309;
      * at the end of data the last model is saved if its eval_dist beats the
        best -- train.py:301-305.

    Unlike the reference (which always re-initialises, train.py:280) a
    checkpoint also carries optimizer slots, step and sampler state, so
    ``resume()`` continues a run exactly."""

    def __init__(self, train_step, num_epochs, n_pairs, checkpoint_dir=None, eval_features=None,
                 eval_cowatches=None, check_stop_epoch=3, best_eval_dist=1.0, eval_per_epoch=100,
                 require_improve_num=10, logger=None, summary_path=None):
        from .evaluate import Evaluation
        from .predict import Prediction
        self.ts = train_step
        B = train_step.batch_global
        self.num_batches = (n_pairs * num_epochs) // B                  # inputs.py:110-122
        self.check_stop_step = int(n_pairs / B * check_stop_epoch)      # train.py:285
        self.step_per_epoch = max(1, int(n_pairs / B))                  # train.py:286
        self.eval_step = max(1, int(n_pairs / B / eval_per_epoch))      # train.py:287
        self.show_step = max(1, int(self.eval_step / 10))               # train.py:288
        self.checkpoint_dir = checkpoint_dir
        self.best_eval_dist = best_eval_dist
        self.eval_dist = 0.0
        self.total_eval_num = 0
        self.last_improve_num = 0
        self.require_improve_num = require_improve_num
        self.log = logger or logging.getLogger("cdml.train")
        self.history, self.eval_history, self.saved = [], [], []
        # the reference's TensorBoard scalars (train.py:154-160, 246-249; losses.py:40-41) as one JSON line per
        # evaluation, under its names: `summary_path` (default <checkpoint_dir>/summaries.jsonl; rank 0 writes)
        if summary_path is None and checkpoint_dir:
            summary_path = os.path.join(checkpoint_dir, "summaries.jsonl")
        self.summary_path = summary_path
        self.summaries = []
        if summary_path and train_step.mode != "semihard" and train_step.var_ws is None:
            train_step.enable_variance()                                # calc_var (train.py:67-71,151) rides in the fused tail
        self.evaluater = None
        if eval_cowatches is not None:
            self.evaluater = Evaluation(eval_features, eval_cowatches, device=train_step.device)
        self.predictor = Prediction(params=train_step.params)

    # ---- checkpoints --------------------------------------------------------
    def save(self, step):
        # never checkpoint weights that were stepped on overflowed (NaN) rows: the check is collective (every rank
        # learns of any rank's overflow and raises here) -- so it comes BEFORE anything that depends on a per-rank
        # argument such as checkpoint_dir (a rank that returned early would leave the others alone in the all-reduce).
        # What is tested is what would be WRITTEN: the weights (and this rank's shard of a trainable table), not the
        # loss of the batch before the update; the verdict rides in the same collective (maximum over the ranks), so
        # every rank takes the same write-or-skip decision and the shard files of one step are all there or none is.
        # The previous checkpoint (max_to_keep = 1 deletes it below) stays the last good one.
        if self.ts.check_inputs(extra=self.ts.weights_nonfinite_flag()):
            self.log.warning("step %d: a weight is not finite (on at least one rank) -- checkpoint NOT written, the "
                             "previous one is kept", step)
            return None
        if not self.checkpoint_dir:
            return None
        # data-parallel runs: the dense state is replicated -> rank 0 writes it; a trainable
        # table is sharded -> every rank writes its own shard file
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        sharded = self.ts.train_table and self.ts.exchange is not None
        if rank != 0 and not sharded:
            return None
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        name = "model.ckpt-%d.rank%d.pt" % (step, rank) if sharded else "model.ckpt-%d.pt" % step
        path = os.path.join(self.checkpoint_dir, name)
        state = self.ts.state_dict()
        state["trainer"] = {"best_eval_dist": self.best_eval_dist, "total_eval_num": self.total_eval_num,
                            "last_improve_num": self.last_improve_num}
        torch.save(state, path)
        for old in self.saved:                                          # max_to_keep=1 (train.py:275)
            if old != path and os.path.exists(old):
                os.remove(old)
        self.saved = [path]
        return path

    def resume(self, path):
        state = torch.load(path, map_location="cpu")
        self.ts.load_state_dict(state)
        for k, v in state.get("trainer", {}).items():
            setattr(self, k, v)

    # ---- evaluation (train.py:224-252) ---------------------------------------
    def _eval(self, global_step):
        self.total_eval_num += 1
        self.ts.check_inputs()
        emb = self.predictor.run_features(self.evaluater.features, batch_size=10000)
        self.eval_dist = self.evaluater.mean_dist(emb, self.evaluater.cowatches)
        if global_step <= self.check_stop_step:
            self.last_improve_num = self.total_eval_num                 # no early stop yet
        elif self.eval_dist < self.best_eval_dist:
            self.best_eval_dist = self.eval_dist
            self.save(global_step)
            self.last_improve_num = self.total_eval_num
        self.eval_history.append((global_step, self.eval_dist, self.best_eval_dist))
        self._emit_summaries(global_step)
        self.log.info("Eval %d | step %d eval_dist %.6f best %.6f", self.total_eval_num, global_step,
                      self.eval_dist, self.best_eval_dist)

    def _emit_summaries(self, global_step):
        """One record per evaluation with the reference's scalar names (train.py:326-327 runs summary_op exactly here)."""
        rec = {"global_step": int(global_step)}
        rec.update(self.ts.summaries())
        rec["eval/eval_dist"] = float(self.eval_dist)                   # train.py:246-249
        rec["eval/best_eval_dist"] = float(self.best_eval_dist)
        self.summaries.append(rec)
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        if self.summary_path and rank == 0:
            import json
            os.makedirs(os.path.dirname(os.path.abspath(self.summary_path)), exist_ok=True)
            with open(self.summary_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    def run(self, max_steps=None):
        n = self.num_batches if max_steps is None else min(self.num_batches, max_steps)
        t0 = time.time()
        stopped = "end of data"
        while self.ts.global_step < n:
            gs = self.ts.global_step
            if (self.total_eval_num - self.last_improve_num > self.require_improve_num
                    and gs > self.check_stop_step):
                stopped = "early stop"
                break
            self.ts.step()
            gs += 1
            if gs % self.show_step == 0 or gs == n:
                loss = self.ts.loss()                   # (also reads the device-side input flags)
                self.history.append((gs, loss))
                self.log.info("Epoch %d Step %d | Loss: %.8f | %.1f triplets/s", gs // self.step_per_epoch + 1,
                              gs, loss, gs * self.ts.batch_global / (time.time() - t0))
            if self.evaluater is not None and gs % self.eval_step == 0:
                self._eval(gs)
        if stopped == "end of data" and self.evaluater is not None and self.eval_dist < self.best_eval_dist:
            self.save(self.ts.global_step)                              # train.py:301-305
        self.stopped = stopped
        return self.history
