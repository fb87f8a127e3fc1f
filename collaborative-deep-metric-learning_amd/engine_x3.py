"""fp32 tower on the bf16 matrix cores (precision "f32x3").

Same layers, same fp32 master weights, loss, normalisation and optimizer kernels as ``engine`` -- only the five
projection GEMMs change: every fp32 operand is held as three bf16 planes hi | mid | lo whose sum IS the fp32
value, and a product is the six plane products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid accumulated in the
fp32 accumulator of ``v_mfma_f32_16x16x32_bf16`` (csrc/gemm_bf16x3.hip: what is dropped is below 2^-26 of a
product).  The bf16 MFMA is sixteen times the fp32 MFMA's rate on gfx950, so six of them for one fp32 product
is 2.7 x faster at equal efficiency.  Errors against fp64 are those of the fp32 kernels
(tests/test_gpu_f32x3.py); the parity tests of the fp32 path run against this one with the same bounds.

Activations never exist in fp32 between the GEMMs: FC1's epilogue writes h1 as planes, the data gradient's
writes dz1 as planes (and takes leaky-relu' from h1's hi plane -- rounding keeps the sign).  Reference lines:
models.py:59-61 (forward), train.py:141 (its autodiff).
"""
import os

import torch

from . import ops
from .engine import TowerLayout, round_up


def layout_x3(feature_size, hidden=5000, output_size=256):
    """TowerLayout whose padded widths are multiples of 256 (the plane GEMMs' tile; 1536 / 5120 / 256 at the
    production sizes, as on the other paths).  Padding rows and columns of the weights are zero and stay zero."""
    L = TowerLayout(feature_size, hidden, output_size)
    L.Fp, L.Hp, L.Dp = round_up(L.F, 256), round_up(L.H, 256), round_up(L.D, 256)
    L.sizes = (L.Fp * L.Hp, L.Hp, L.Hp * L.Dp, L.Dp)
    off = [0]
    for n in L.sizes[:-1]:
        off.append(off[-1] + n)
    L.offsets = tuple(off)
    L.numel = int(sum(L.sizes))
    return L


class TowerWorkspaceX3:
    def __init__(self, layout, n_rows, device, products=6, planes_in=True, backward=True, transposed=None,
                 fc2_single_pass=False, kint=None):
        """planes_in: ``x_hat`` IS the plane buffer (the fused sampler + gather writes planes); False: ``x_hat`` is
        fp32 (rows arriving through the exchange) and the forward pass splits it.
        fc2_single_pass: the narrow second layer normally splits its contraction into slabs whose partition depends on
        K alone, so an embedding has the same bits whatever batch or chunk it was computed in (whole batch or row
        blocks, 10 000-row evaluation chunks or 65 536-row inference chunks).  True = the explicit opt-out for forward-only
        workspaces of >= 192 row tiles: ONE pass over K, no slab round trip (about 5 % of an inference chunk), last bits
        that differ from the slab form's.
        kint (training, six products, planes_in; OFF unless CDML_X3_KI=1 -- measured, see the end): the first layer's weight gradient
        contracts over the batch rows, so its operands -- x_hat and dz1 -- are k-STRIDED in their row-major form and every
        fragment costs two transposed LDS reads.  With kint they are ALSO (x_hat: the fused gather writes a second copy,
        ``xk``) or ONLY (dz1: the data gradient's epilogue writes ``dz1k`` instead of ``dz1``) held k8-interleaved --
        [plane][row / 8][column][8 rows] -- and dW1 runs on cdml_gemm_bf16x3_tnk: one aligned 16-B LDS read per fragment, the
        same images, DMA schedule, accumulation order and BITS (tests/test_gpu_f32x3.py).  ``xk`` is set by the owner of
        the gather buffers (TrainStep); without it the backward pass falls back to the row-major product.
        MEASURED (profiles/r05_tnk_probe.txt, r05_kint_in_the_step.txt): alone, back to back, dW1 runs 10-12 % faster this way
        (0.59 -> 0.655 of the peak at 16 384 rows); IN THE STEP 4 % (999 -> 959 us: the chip answers the higher matrix-pipe
        duty with a lower clock), the gather's second copy costs + 48 us a step (151 MB more to write and a launch per step
        instead of per two), the epilogue's second LDS pass + 5 us: the step comes out EVEN (2.7309 against 2.7321 ms).  Kept,
        with its tests, as the measured alternative; off by default."""
        L, R = layout, int(n_rows)
        if R % 128:
            raise ValueError("precision 'f32x3' needs a row count that is a multiple of 128 (got %d)" % R)
        if products not in (3, 6):
            raise ValueError("products must be 6 (fp32-equivalent) or 3 (16-bit operands)")
        self.layout, self.R, self.products = L, R, products
        bf = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=device)
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        # ``transposed`` (training, R % 256 == 0, six products): the hidden layer's activations and their gradient are
        # held TRANSPOSED -- h1^T, dz1^T as [H][hi R | mid R | lo R] -- by computing FC1 and the data gradient with their
        # operands swapped (C^T = W^T-planes . x-planes^T).  Every product but the narrow FC2 then has BOTH operands
        # k-contiguous: the weight gradients, which contract over the batch rows, run in the k-contiguous form (0.57 of
        # the bf16 peak / 6 against 0.47 for the transposed LDS reads of the k-strided form); FC2 (6 % of the flop) takes
        # the k-strided form instead, and the gathered planes get one transposed copy.
        # MEASURED SLOWER (1.69 against 1.62 ms/step on one box): dW1 gains less than hoped (605 -> 558 us: it needs a 2-way
        # K split with a slab combine where FC1 does not, and the general loop -- the unrolled period plus the column sums
        # does not fit 256 VGPRs), dW2 gains 7 us, FC2 loses 13, and the transposed copy
        # of the gathered planes costs 88 us in three launches -- even a perfect copy kernel (25 us) would only draw
        # level.  Off unless CDML_X3_TRANSPOSED=1; kept, with its tests, as the measured alternative.
        if transposed is None:
            transposed = (os.environ.get("CDML_X3_TRANSPOSED") == "1" and backward and R % 256 == 0 and products == 6)
        if transposed and (R % 256 or not backward):
            raise ValueError("the transposed activation layout needs a training workspace with rows % 256 == 0")
        self.transposed = bool(transposed)
        maskbits = backward and not self.transposed and os.environ.get("CDML_X3_MASKBITS", "1") != "0"
        if kint is None:
            kint = os.environ.get("CDML_X3_KI", "0") == "1" and planes_in
        self.kint = bool(kint) and maskbits and products == 6 and R % 8 == 0
        self.xk = None                                       # [3 * R * Fp] bf16, k8-interleaved x_hat (the gather's second output)
        self.x3 = bf(R, 3 * L.Fp)
        self.x_hat = self.x3 if planes_in else f32(R, L.Fp)    # the gather's output (l2-normalised rows)
        self.h1 = bf(L.Hp, 3 * R) if self.transposed else bf(R, 3 * L.Hp)
        self.z, self.e = f32(R, L.Dp), f32(R, L.Dp)
        self.W1T, self.W2T, self.W2 = bf(L.Hp, 3 * L.Fp), bf(L.Dp, 3 * L.Hp), bf(L.Hp, 3 * L.Dp)
        q = products
        # FC2 (one tile column) splits its contraction into slabs -- a partition that depends on K alone, so a batch gives
        # the same bits whole or in row blocks, in small chunks or large (test_embedding_bits_do_not_depend_on_the_chunk);
        # the single pass (workspace NULL at the C ABI) is an explicit opt-out, fc2_single_pass.
        self.fc2_single_pass = bool(fc2_single_pass) and not backward and R // 256 >= 192
        # round 6: the slab LENGTH follows the row-tile class of a call (60 K-tile steps for small batches such as the
        # reference's own B = 1 024, 120 otherwise: csrc/gemm_bf16x3.hip x3_nt_splits).  A forward-only workspace -- catalogue
        # inference, evaluation -- pins 120 whatever its chunk size, so an embedding keeps its bits from chunk to chunk; a
        # training workspace follows the rule (its batch size is what it is: one class per job)
        self.slab_steps = None if backward else 120
        nb = 16 if self.fc2_single_pass else max(ops.gemm_bf16x3_workspace(False, R, L.Dp, L.Hp, q), 16)
        # leaky-relu' of the hidden layer as ONE BIT per element (round 4): FC1's epilogue writes the sign bitmask of h1
        # (this lane's 8 columns = one byte), the data gradient's epilogue reads 5 MB of bits instead of the 84 MB of
        # h1's hi plane -- that read sat in its store-bound epilogue and cost 30 us of its 151 (profiles/r04_stagger_and_
        # fc1_rounds.txt, item 3).  CDML_X3_MASKBITS=0 for the value mask (A/B runs); the transposed layout keeps it.
        self.h1_bits = None
        if maskbits:
            self.h1_bits = torch.zeros((R, L.Hp // 8), dtype=torch.uint8, device=device)
        if backward:                                       # (catalogue inference: forward buffers only)
            self.dz1 = bf(L.Hp, 3 * R) if self.transposed else (None if self.kint else bf(R, 3 * L.Hp))
            self.dz1k = torch.zeros(3 * R * L.Hp, dtype=torch.bfloat16, device=device) if self.kint else None
            self.de, self.dz2 = f32(R, L.Dp), f32(R, L.Dp)
            self.dz2_3 = bf(R, 3 * L.Dp)
            nb = max(nb, ops.gemm_bf16x3_workspace(True, L.Fp, L.Hp, R, q), ops.gemm_bf16x3_workspace(True, L.Hp, L.Dp, R, q))
            if L.Fp % 512 == 0:                            # the bucketed data-parallel step: dW1 in two row blocks of W1,
                nb = max(nb, ops.gemm_bf16x3_workspace(True, L.Fp // 2, L.Hp, R, q))   # which may pick more slabs each
            if self.transposed:
                self.xT = bf(L.Fp, 3 * R)                  # the gathered planes, transposed (dW1's A operand)
                self.dz2T = bf(L.Dp, 3 * R)                # dz2's planes, transposed (dW2's B operand)
                nb = max(nb, ops.gemm_bf16x3_workspace(True, R, L.Dp, L.Hp, q),        # FC2 in the k-strided form
                         ops.gemm_bf16x3_workspace(False, L.Fp, L.Hp, R, q), ops.gemm_bf16x3_workspace(False, L.Hp, L.Dp, R, q))
                if L.Fp % 512 == 0:
                    nb = max(nb, ops.gemm_bf16x3_workspace(False, L.Fp // 2, L.Hp, R, q))
        self.gemm_ws = torch.empty(nb // 4, dtype=torch.float32, device=device)
        self.tail_done = False

    def _sum_planes(self, t, width):
        return t[:, :width].float() + t[:, width:2 * width].float() + t[:, 2 * width:3 * width].float()

    def x_hat_f32(self):
        """the gathered, l2-normalised rows as one fp32 tensor [R, Fp] (tests, debugging)"""
        return self.x_hat if self.x_hat.dtype == torch.float32 else self._sum_planes(self.x_hat, self.layout.Fp)

    def dz1_f32(self):
        """the hidden layer's pre-activation gradient as one fp32 tensor [R, Hp]"""
        if self.transposed:
            return self._sum_planes(self.dz1, self.R).t().contiguous()
        if self.dz1 is None:                               # k8-interleaved: [3][R / 8][Hp][8]
            v = self.dz1k.view(3, self.R // 8, self.layout.Hp, 8).permute(0, 1, 3, 2).reshape(3, self.R, self.layout.Hp)
            return v[0].float() + v[1].float() + v[2].float()
        return self._sum_planes(self.dz1, self.layout.Hp)

    def h1_f32(self):
        """the hidden activations as one fp32 tensor (tests, debugging): the planes summed"""
        H, R = self.layout.Hp, self.R
        if self.transposed:
            return (self.h1[:, :R].float() + self.h1[:, R:2 * R].float() + self.h1[:, 2 * R:].float()).t().contiguous()
        return self.h1[:, :H].float() + self.h1[:, H:2 * H].float() + self.h1[:, 2 * H:].float()


def refresh_weights(p, ws):
    """plane copies of the fp32 master weights in the orientations the GEMMs read (after every optimizer step)"""
    L = p.layout
    ops.split_f32_bf16x3(p.W1, ws.W1T, L.Fp, transpose=True)        # [Hp][3 Fp]
    ops.split_f32_bf16x3(p.W2, ws.W2T, L.Hp, transpose=True)        # [Dp][3 Hp]
    ops.split_f32_bf16x3(p.W2, ws.W2, L.Dp)                         # [Hp][3 Dp]
    if getattr(ws, "W1n", None) is not None:                        # trainable catalogue: dx_hat = dz1 . W1^T reads W1 as it is
        ops.split_f32_bf16x3(p.W1, ws.W1n, L.Hp)                    # [Fp][3 Hp]


def tower_forward(p, ws, normalize=True):
    """x_hat (fp32, l2-normalised) -> planes -> h1 (planes) -> z (fp32) -> e (fp32).  models.py:59-61."""
    L, R, q = p.layout, ws.R, ws.products
    if ws.x_hat.dtype == torch.float32:                 # rows from the exchange / a caller: split here; the fused
        ops.split_f32_bf16x3(ws.x_hat, ws.x3, L.Fp)     # sampler + gather writes the planes itself (x_hat IS x3 then)
    else:
        ws.x3 = ws.x_hat
    if ws.transposed:
        # h1^T = lrelu(W1^T x^T + b1[row]): the weights as the row operand; then z = (h1^T)^T W2 in the k-strided form
        ops.gemm_bf16x3_nt(ops.BE_ROWBIAS_LRELU_X3, ws.W1T, L.Fp, ws.x3, L.Fp, ws.h1, L.Hp, R, L.Fp, products=q,
                           plane_c=R, bias=p.b1)
        ops.gemm_bf16x3_tn(ws.h1, R, ws.W2, L.Dp, ws.z, R, L.Dp, L.Hp, products=q, workspace=ws.gemm_ws, bias=p.b2)
    else:
        if getattr(ws, "h1_bits", None) is not None:
            ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3_BITS, ws.x3, L.Fp, ws.W1T, L.Fp, ws.h1, R, L.Hp, L.Fp, products=q,
                               plane_c=L.Hp, bias=p.b1, aux=ws.h1_bits)
        else:
            ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, ws.x3, L.Fp, ws.W1T, L.Fp, ws.h1, R, L.Hp, L.Fp, products=q,
                               plane_c=L.Hp, bias=p.b1)
        ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, ws.h1, L.Hp, ws.W2T, L.Hp, ws.z, R, L.Dp, L.Hp, products=q,
                           bias=p.b2, workspace=None if getattr(ws, "fc2_single_pass", False) else ws.gemm_ws,
                           slab_steps=getattr(ws, "slab_steps", None))
    ws.tail_done = False
    ws.dz2_planes_done = False
    if normalize:
        ops.l2norm_fwd(ws.z, L.Dp, ws.e)
    return ws.e


def tower_backward(p, ws, after_w1=None, w1_chunks=1, after_w1_chunk=None):
    """ws.dz2 (from the fused tail) or ws.de -> p.grad (fp32).  train.py:141; no dX.  The second layer's weight
    gradient goes first on one GPU (it and the data gradient both stream h1); with ``after_w1`` (the
    data-parallel all-reduce of [dW1|db1]) the first layer's goes first and the hook fires right after it; with
    ``w1_chunks`` > 1 dW1 comes in row blocks of W1 and ``after_w1_chunk(lo, hi)`` fires after each (flat-gradient
    ranges; the last one ends after db1), as in engine.tower_backward."""
    L, R, q = p.layout, ws.R, ws.products
    if not ws.tail_done:
        ops.l2norm_bwd(ws.z, ws.de, L.Dp, ws.dz2, lrelu_alpha=ops.LRELU_ALPHA)
    if not (ws.tail_done and getattr(ws, "dz2_planes_done", False)):      # the fused tail writes the planes itself
        ops.split_f32_bf16x3(ws.dz2, ws.dz2_3, L.Dp)
    T = ws.transposed
    if T:
        ops.split_f32_bf16x3(ws.dz2, ws.dz2T, R, transpose=True)                  # [Dp][3 R]
        # both operands k-contiguous (k = the batch rows): dW2 = h1^T . (dz2^T)^T, db2 = the row sums of dz2^T
        w2 = lambda: ops.gemm_bf16x3_nt(ops.BE_F32, ws.h1, R, ws.dz2T, R, p.gW2, L.Hp, L.Dp, R, products=q,
                                        workspace=ws.gemm_ws, colsum=p.gb2)
    else:
        w2 = lambda: ops.gemm_bf16x3_tn(ws.h1, L.Hp, ws.dz2_3, L.Dp, p.gW2, L.Hp, L.Dp, R, products=q,
                                        workspace=ws.gemm_ws, colsum=p.gb2)
    single = after_w1 is None and after_w1_chunk is None
    if single:
        w2()
    # dz1 = (dz2 . W2^T) * lrelu'(h1), written as planes; the sign comes from h1's hi plane
    if T:
        ops.gemm_bf16x3_nt(ops.BE_MASK_X3, ws.W2, L.Dp, ws.dz2_3, L.Dp, ws.dz1, L.Hp, R, L.Dp, products=q, plane_c=R,
                           aux=ws.h1)                                             # dz1^T = (W2 dz2^T) * lrelu'(h1^T)
        for pl in range(3):                                                       # the gathered planes, transposed
            ops.transpose_to_bf16(ws.x3[:, pl * L.Fp:(pl + 1) * L.Fp], ws.xT[:, pl * R:(pl + 1) * R], R, L.Fp)
    else:
        kint = getattr(ws, "kint", False) and ws.xk is not None and getattr(ws, "h1_bits", None) is not None
        if getattr(ws, "kint", False) and not kint:
            raise RuntimeError("this workspace holds dz1 k8-interleaved only: it needs the gather's interleaved x_hat (ws.xk) and "
                               "the sign bitmask of h1 (CDML_X3_MASKBITS); build it with kint=False otherwise")
        if kint:      # dz1 written k8-interleaved by the epilogue: the only form the first layer's weight gradient reads
            ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3_KI, ws.dz2_3, L.Dp, ws.W2, L.Dp, ws.dz1k, R, L.Hp, L.Dp, products=q,
                               plane_c=R * L.Hp, aux=ws.h1_bits, ldc=L.Hp)
        elif getattr(ws, "h1_bits", None) is not None:
            ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3, ws.dz2_3, L.Dp, ws.W2, L.Dp, ws.dz1, R, L.Hp, L.Dp, products=q,
                               plane_c=L.Hp, aux=ws.h1_bits)
        else:
            ops.gemm_bf16x3_nt(ops.BE_MASK_X3, ws.dz2_3, L.Dp, ws.W2, L.Dp, ws.dz1, R, L.Hp, L.Dp, products=q, plane_c=L.Hp,
                               aux=ws.h1)
    rows = L.Fp // w1_chunks if w1_chunks > 1 else 0

    def dw1(lo, hi, db):
        if T:      # rows lo .. hi of x^T against all of dz1^T; db1 = the row sums of dz1^T
            ops.gemm_bf16x3_nt(ops.BE_F32, ws.xT[lo:hi], R, ws.dz1, R, p.gW1[lo:hi], hi - lo, L.Hp, R, products=q,
                               workspace=ws.gemm_ws, colsum=db)
        elif getattr(ws, "kint", False):      # k8-interleaved operands: one 16-B LDS read per fragment (columns lo .. hi of x_hat)
            ops.gemm_bf16x3_tnk(ws.xk, L.Fp, lo, ws.dz1k, L.Hp, 0, p.gW1[lo:hi], hi - lo, L.Hp, R, workspace=ws.gemm_ws, colsum=db)
        else:      # columns lo .. hi of every plane of x_hat: the same plane stride, the base moved by lo
            ops.gemm_bf16x3_tn(ws.x3[:, lo:], L.Fp, ws.dz1, L.Hp, p.gW1[lo:hi], hi - lo, L.Hp, R, products=q,
                               workspace=ws.gemm_ws, colsum=db)
    if after_w1_chunk is not None and w1_chunks > 1 and rows * w1_chunks == L.Fp and rows % 256 == 0:
        for c in range(w1_chunks):
            last = c == w1_chunks - 1
            dw1(c * rows, (c + 1) * rows, p.gb1 if last else None)
            after_w1_chunk(c * rows * L.Hp, (c + 1) * rows * L.Hp + (L.Hp if last else 0))
    else:
        dw1(0, L.Fp, p.gb1)
        if after_w1_chunk is not None:
            after_w1_chunk(0, L.Fp * L.Hp + L.Hp)
    if after_w1 is not None:
        after_w1()
    if not single:
        w2()
    return p.grad
