"""Reduced-precision tower for BASELINE config 4: fp16 catalogue, bf16 MFMA
projection, fp32 accumulation, fp32 master weights and optimizer.

Build-defined precision (the reference computes in fp32): same layers and the
same fp32 loss / normalisation kernels as ``engine``, but the four projection
GEMMs and the data gradient run on ``v_mfma_f32_32x32x16_bf16`` through the one
k-contiguous form ``C = A . B^T`` (``ops.gemm_bf16_nt``).  Operands that are
k-strided in memory (the weight gradients contract over batch rows) are fed
from transposed bf16 copies.  Tolerance stated in the tests: 5e-3 absolute on the
unit-norm embeddings, 2e-2 on the loss.
"""
import os

import torch

from . import ops
from .engine import FeatureTable, TowerLayout, round_up


class FeatureTableF16(FeatureTable):
    """fp16 catalogue shard [n_rows, row_stride] (3 KB rows at F=1500)."""

    def __init__(self, data, feature_size, row0=0, n_rows_global=None):
        if not data.is_cuda or data.dtype != torch.float16 or data.dim() != 2:
            raise ValueError("FeatureTableF16 needs a 2-D fp16 device tensor")
        self.data = data
        self.feature_size = int(feature_size)
        self.row0 = int(row0)
        self.n_rows = data.shape[0]
        self.n_rows_global = self.n_rows if n_rows_global is None else int(n_rows_global)

    @classmethod
    def from_numpy(cls, features, device, row0=0, n_rows_global=None):
        import numpy as np
        f = np.asarray(features, dtype=np.float16)
        n, F = f.shape
        data = torch.zeros((n, cls.padded_stride(F)), dtype=torch.float16, device=device)
        data[:, :F] = torch.from_numpy(f).to(device)
        return cls(data, F, row0, n_rows_global)

    @classmethod
    def synthetic(cls, n_rows, feature_size, seed, device, row0=0, n_rows_global=None):
        data = torch.empty((n_rows, cls.padded_stride(feature_size)), dtype=torch.float16, device=device)
        ops.fill_uniform_table_f16(data, row0, feature_size, seed)
        return cls(data, feature_size, row0, n_rows_global)


def layout_bf16(feature_size, hidden=5000, output_size=256):
    """TowerLayout whose padded sizes satisfy the bf16 GEMM (N % 128, K % 64)."""
    L = TowerLayout(feature_size, hidden, output_size)
    if L.Dp % 128:
        L.Dp = round_up(L.D, 128)
        L.sizes = (L.Fp * L.Hp, L.Hp, L.Hp * L.Dp, L.Dp)
        off = [0]
        for n in L.sizes[:-1]:
            off.append(off[-1] + n)
        L.offsets = tuple(off)
        L.numel = int(sum(L.sizes))
    return L


class TowerWorkspaceBF16:
    def __init__(self, layout, n_rows, device, backward=True):
        L, R = layout, int(n_rows)
        if R % 64:
            raise ValueError("the bf16 path needs a row count that is a multiple of 64 (got %d)" % R)
        self.layout, self.R = L, R
        bf = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=device)
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.x_hat, self.h1 = bf(R, L.Fp), bf(R, L.Hp)
        self.z, self.e = f32(R, L.Dp), f32(R, L.Dp)
        self.W1T, self.W2T, self.W2 = bf(L.Hp, L.Fp), bf(L.Dp, L.Hp), bf(L.Hp, L.Dp)
        if not backward:                                   # catalogue inference: forward buffers only
            self.gemm_ws = torch.empty(max(ops.gemm_bf16_workspace(R, L.Dp, L.Hp), 16) // 4, dtype=torch.float32,
                                       device=device)
            return
        # weight gradients straight from the activations as stored (k-strided GEMM with
        # transposed LDS reads) when the shapes allow; otherwise transposed bf16 copies
        self.tn1 = ops.gemm_bf16_tn_supported(L.Fp, L.Hp, R, L.Fp, L.Hp)
        self.tn2 = ops.gemm_bf16_tn_supported(L.Hp, L.Dp, R, L.Hp, L.Dp)
        self.dz1 = bf(R, L.Hp)
        # leaky-relu' of the hidden layer as ONE BIT per element, written by FC1's epilogue and read by
        # the data gradient instead of the 2-byte activations (epilogues 4 / 5).  OFF by default: measured
        # on one box in the config-4 step (profiles/r03_bf16_maskbits_ab.txt) the byte stores cost FC1's
        # epilogue 23 us and the data gradient did not get faster -- it is not bound by the mask bytes
        # (tools/experiments/k256_ablate.sh: without any mask traffic 86 us of its 119)
        self.h1_bits = None
        if (os.environ.get("CDML_BF16_MASKBITS") == "1"
                and ops.gemm_bf16_epilogue_supported(ops.BE_BIAS_LRELU_BF16_BITS, R, L.Hp, L.Fp, L.Fp, L.Fp, L.Hp, L.Hp // 8)
                and ops.gemm_bf16_epilogue_supported(ops.BE_MASKBITS_BF16, R, L.Hp, L.Dp, L.Dp, L.Dp, L.Hp, L.Hp // 8)):
            self.h1_bits = torch.zeros((R, L.Hp // 8), dtype=torch.uint8, device=device)
        self.de, self.dz2 = f32(R, L.Dp), f32(R, L.Dp)
        self.dz2_bf = bf(R, L.Dp)
        if not self.tn1:
            self.xT, self.dz1T = bf(L.Fp, R), bf(L.Hp, R)
        if not self.tn2:
            self.h1T, self.dz2T = bf(L.Hp, R), bf(L.Dp, R)
        # both weight gradients in one stream-K launch when the shapes allow (0 = they do not)
        self.tn2_bytes = ops.gemm_bf16_tn2_workspace(L.Fp, L.Hp, L.Hp, L.Dp, R) if (self.tn1 and self.tn2) else 0
        nb = max(self.tn2_bytes,
                 ops.gemm_bf16_workspace(L.Hp, L.Dp, R), ops.gemm_bf16_workspace(L.Fp, L.Hp, R),
                 ops.gemm_bf16_workspace(R, L.Dp, L.Hp),
                 ops.gemm_bf16_tn_workspace(L.Hp, L.Dp, R), ops.gemm_bf16_tn_workspace(L.Fp, L.Hp, R),
                 ops.gemm_bf16_tn_workspace(L.Fp // 2, L.Hp, R), 16)      # dW1 in two row blocks (data-parallel)
        self.gemm_ws = torch.empty(nb // 4, dtype=torch.float32, device=device)
        self.colsum_ws = f32(max(ops.colsum_workspace_floats(R, L.Hp), ops.colsum_workspace_floats(R, L.Dp)))


def refresh_weights(p, ws):
    """bf16 operand copies of the fp32 master weights (after every optimizer step)."""
    L = p.layout
    ops.transpose_to_bf16(p.W1, ws.W1T, L.Fp, L.Hp)
    ops.transpose_to_bf16(p.W2, ws.W2T, L.Hp, L.Dp)
    ops.cast_f32_bf16(p.W2, ws.W2, L.Hp, L.Dp)


def tower_forward(p, ws, normalize=True):
    """x_hat (bf16, l2-normalised) -> h1 (bf16) -> z (fp32) -> e (fp32).  models.py:59-61.
    ``normalize=False``: stop at z (the fused tail of the training step takes over)."""
    L, R = p.layout, ws.R
    bits = getattr(ws, "h1_bits", None)
    if bits is not None:
        ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16_BITS, ws.x_hat, ws.W1T, ws.h1, R, L.Hp, L.Fp, bias=p.b1, aux=bits)
    else:
        ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16, ws.x_hat, ws.W1T, ws.h1, R, L.Hp, L.Fp, bias=p.b1)
    ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_F32, ws.h1, ws.W2T, ws.z, R, L.Dp, L.Hp, bias=p.b2, workspace=ws.gemm_ws)
    ws.tail_done = False
    if normalize:
        ops.l2norm_fwd(ws.z, L.Dp, ws.e)
    return ws.e


def tower_backward(p, ws, after_w1=None, w1_chunks=1, after_w1_chunk=None):
    """ws.de -> p.grad (fp32).  train.py:141; no dX.  As in the fp32 engine the first layer's
    gradient (85 % of the bytes) is produced BEFORE the second layer's, so that ``after_w1`` --
    the data-parallel all-reduce of [dW1|db1] -- runs under the dW2 GEMM and the db2 sums;
    with ``w1_chunks`` > 1 dW1 comes in row blocks of W1 and ``after_w1_chunk(lo, hi)`` fires
    after each (flat-gradient ranges; the last one ends after db1), as in engine.tower_backward.
    Without those hooks (one GPU) the second layer's gradient goes FIRST: it and the data gradient
    that follows both stream h1 (252 MB at the config-4 shape), and back to back the second pass
    finds part of it in the Infinity Cache (1.087 -> 1.070 ms per step, measured)."""
    L, R = p.layout, ws.R
    if not getattr(ws, "tail_done", False):          # the fused tail writes dz2 and its bf16 copy
        ops.l2norm_bwd(ws.z, ws.de, L.Dp, ws.dz2, lrelu_alpha=ops.LRELU_ALPHA)
        ops.cast_f32_bf16(ws.dz2, ws.dz2_bf, R, L.Dp)
    single = after_w1 is None and after_w1_chunk is None
    how = os.environ.get("CDML_BF16_DW", "split")                        # (the switch: A/B runs)
    joint = single and ws.tn2_bytes and how == "joint"
    w2_first = single and not joint and ws.tn2 and not os.environ.get("CDML_BF16_W2_LAST")
    if joint:
        # NOT the default (profiles/r03_bf16_joint_dw.txt): data gradient first, then BOTH weight gradients and
        # both bias gradients in one stream-K launch + fix-up pass.  Every block gets the same number of K-tiles,
        # but blocks that sit at different k share no operand panels in L2 and the launch runs 29 % slower than
        # the two split-K launches (the same finding as the fp32 pure stream-K of round 2).  Running dW2 on a
        # side stream beside dW1 was measured too: the two finish 23 us sooner, the fork/join costs 34.
        if ws.h1_bits is not None:
            ops.gemm_bf16_nt(ops.BE_MASKBITS_BF16, ws.dz2_bf, ws.W2, ws.dz1, R, L.Hp, L.Dp, aux=ws.h1_bits)
        else:
            ops.gemm_bf16_nt(ops.BE_MASK_BF16, ws.dz2_bf, ws.W2, ws.dz1, R, L.Hp, L.Dp, aux=ws.h1)
        ops.gemm_bf16_tn2(ws.x_hat, ws.dz1, p.gW1, L.Fp, L.Hp, ws.h1, ws.dz2_bf, p.gW2, L.Hp, L.Dp, R, ws.gemm_ws,
                          colsum1=p.gb1, colsum2=p.gb2)
        return p.grad
    if w2_first:
        # db2 = column sums of the bf16 dz2 the two products consume, from the LDS tiles of the same GEMM
        ops.gemm_bf16_tn(ws.h1, ws.dz2_bf, p.gW2, L.Hp, L.Dp, R, workspace=ws.gemm_ws, colsum=p.gb2)
    if ws.h1_bits is not None:
        ops.gemm_bf16_nt(ops.BE_MASKBITS_BF16, ws.dz2_bf, ws.W2, ws.dz1, R, L.Hp, L.Dp, aux=ws.h1_bits)
    else:
        ops.gemm_bf16_nt(ops.BE_MASK_BF16, ws.dz2_bf, ws.W2, ws.dz1, R, L.Hp, L.Dp, aux=ws.h1)
    rows = L.Fp // w1_chunks if w1_chunks > 1 else 0
    chunked = (after_w1_chunk is not None and w1_chunks > 1 and rows * w1_chunks == L.Fp and ws.tn1
               and ops.gemm_bf16_tn_supported(rows, L.Hp, R, L.Fp, L.Hp))
    if chunked:
        for c in range(w1_chunks):
            lo, hi = c * rows, (c + 1) * rows
            last = c == w1_chunks - 1
            ops.gemm_bf16_tn(ws.x_hat[:, lo:hi], ws.dz1, p.gW1[lo:hi], rows, L.Hp, R, workspace=ws.gemm_ws,
                             colsum=p.gb1 if last else None)
            after_w1_chunk(lo * L.Hp, hi * L.Hp + (L.Hp if last else 0))
    else:
        if ws.tn1:   # db1 = column sums of dz1, taken from the LDS tiles of the same GEMM
            ops.gemm_bf16_tn(ws.x_hat, ws.dz1, p.gW1, L.Fp, L.Hp, R, workspace=ws.gemm_ws, colsum=p.gb1)
        else:
            ops.colsum(ws.dz1, R, L.Hp, p.gb1, ws.colsum_ws)
            ops.transpose_to_bf16(ws.dz1, ws.dz1T, R, L.Hp)
            ops.transpose_to_bf16(ws.x_hat, ws.xT, R, L.Fp)
            ops.gemm_bf16_nt(ops.BE_F32, ws.xT, ws.dz1T, p.gW1, L.Fp, L.Hp, R, workspace=ws.gemm_ws)
        if after_w1_chunk is not None:
            after_w1_chunk(0, L.Fp * L.Hp + L.Hp)
    if after_w1 is not None:
        after_w1()
    if w2_first:
        return p.grad
    if ws.tn2:
        ops.gemm_bf16_tn(ws.h1, ws.dz2_bf, p.gW2, L.Hp, L.Dp, R, workspace=ws.gemm_ws, colsum=p.gb2)
    else:
        ops.colsum(ws.dz2, R, L.Dp, p.gb2, ws.colsum_ws)
        ops.transpose_to_bf16(ws.dz2, ws.dz2T, R, L.Dp)
        ops.transpose_to_bf16(ws.h1, ws.h1T, R, L.Hp)
        ops.gemm_bf16_nt(ops.BE_F32, ws.h1T, ws.dz2T, p.gW2, L.Hp, L.Dp, R, workspace=ws.gemm_ws)
    return p.grad
