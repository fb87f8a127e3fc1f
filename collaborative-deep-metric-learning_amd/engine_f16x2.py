"""fp32 tower on the fp16 matrix cores (precision "f16x2"; round 6).

Same layers, same fp32 master weights, loss, normalisation and optimizer arithmetic as ``engine`` / ``engine_x3`` -- only
the operand form of the five projection GEMMs changes.  Precision "f32x3" holds an fp32 value as three exact bf16 planes
and multiplies six plane products; here a tensor x is held as TWO fp16 planes

    x 2^s = hi + lo,   hi = fp16(x 2^s),   lo = fp16(x 2^s - hi)            (s: a power of two per tensor, exact)

and a product is the THREE plane products hi.hi + hi.lo + lo.hi on ``v_mfma_f32_16x16x32_f16`` in one fp32 accumulator,
times 2^-(sa + sb): half the matrix work (csrc/gemm_f16x2_256.hip: the five products of the headline step 1.57 x faster,
profiles/r06_f16x2_kernels_rate.txt).  What it costs:

* PRECISION.  hi + lo holds 22 of the 24 significant bits and lo.lo is dropped: below 2^-22 of a product -- under the
  fp32-MFMA kernel's own accumulation error at every shape of the tower (tests/test_gpu_f16x2.py: the unchanged bound of
  precision "f32x3", error against fp64 <= 1.5 x the fp32-MFMA kernel's; the probe that decided it,
  profiles/r06_f16x2_probe.txt).
* RANGE.  bf16 has fp32's exponent; fp16 has 40 binades, of which a plane pair keeps full relative accuracy over about 18
  below its largest value.  Hence the scales: every plane tensor is written times a power of two that puts its largest
  magnitude near 2^11 .. 2^13 -- x_hat's is a constant (|x_hat| <= 1: 2^14), the weights' follow their maxima, h1's and
  dz1's follow RIGOROUS bounds (|h1| <= max column norm of W1 + max |b1|; |dz1| <= max row norm of dz2 x max row norm of
  W2), dz2's its maximum.  The scales live on the HOST (they are kernel arguments): ``PlaneScales`` re-derives them at
  steps 0, 1, 2, 4, ... 64 and every 64th step after (two small device-to-host copies each), and changes one only when its
  tensor has left a window of 2^8 around its place -- delayed scaling, as fp8 training does.  Between two checks a tensor
  may grow 8 .. 32 x before anything saturates, and what saturates is clamped at +-65504, not turned into infinities -- every
  check also looks for that value in the hi planes of h1 and dz1 and counts / logs it (``PlaneScales.saturated``).
  This is an fp32 EQUIVALENT only while that holds: a secondary path (the headline stays "f32x3"), eager (a scale is an argument
  baked into a captured graph).  Data-parallel runs need no agreement on the scales: the weights are replicated (same maxima on
  every rank), the gradient scales are local to a rank's batch, and what crosses the wire -- rows, fp32 gradients -- carries none.

Reference lines: models.py:59-61 (forward), train.py:141 (its autodiff)."""
import math

import torch

from . import ops
from .engine_x3 import layout_x3                     # the same padded widths: multiples of 256

X_SCALE = 2.0 ** 14                                  # include/cdml.h CDML_F16X2_X_SCALE: the gather's constant


def pow2_for(amax, top):
    """the power of two that puts ``amax`` in (top / 2, top]; exponents kept where their products stay finite in fp32"""
    if not (amax > 0.0) or math.isinf(amax) or math.isnan(amax):
        return None
    e = math.floor(math.log2(top / amax))
    return 2.0 ** max(-30, min(40, e))


class PlaneScales:
    """The per-tensor powers of two of a "f16x2" workspace and the rule that moves them (module docstring)."""
    TOP_MAX, TOP_BOUND = 2.0 ** 11, 2.0 ** 13        # where a maximum / a rigorous bound is put
    WINDOW = 2.0 ** 8                                # a scale stays while its tensor is within [top / WINDOW, 4 top)

    def __init__(self, check_every=64):
        self.x = X_SCALE
        self.w1 = self.w2 = self.h1 = self.dz2 = self.dz1 = 1.0
        self.calibrated = False
        self.check_every = int(check_every)
        self.changes = 0                             # how many times a scale moved after calibration
        self.saturated = 0                           # checks that found a plane tensor AT fp16's largest value (it should never)
        self.last = {}                               # the maxima / bounds of the last check (host floats)

    def due(self, step):
        return (not self.calibrated) or step in (0, 1, 2, 4, 8, 16, 32) or step % self.check_every == 0

    def _move(self, name, value, top):
        """value = the tensor's maximum (or bound) as of now; returns True when the scale changed"""
        cur = getattr(self, name)
        new = pow2_for(value, top)
        if new is None:
            return False
        v = value * cur
        if self.calibrated and top / self.WINDOW <= v < 4.0 * top:
            return False
        if new == cur:
            return False
        setattr(self, name, new)
        if self.calibrated:
            self.changes += 1
        return True

    def state(self):
        return {k: getattr(self, k) for k in ("x", "w1", "w2", "h1", "dz2", "dz1")}


class TowerWorkspaceH2:
    def __init__(self, layout, n_rows, device, planes_in=True, backward=True, check_every=64):
        """planes_in: ``x_hat`` IS the plane buffer (the fused sampler + gather writes the fp16 planes); False: ``x_hat``
        is fp32 and the forward pass splits it."""
        L, R = layout, int(n_rows)
        if R % 128:
            raise ValueError("precision 'f16x2' needs a row count that is a multiple of 128 (got %d)" % R)
        if L.Fp % 128 or L.Hp % 128 or L.Dp % 128:
            raise ValueError("precision 'f16x2' needs padded widths that are multiples of 128 (layout_x3)")
        self.layout, self.R = L, R
        h = lambda *s: torch.zeros(s, dtype=torch.float16, device=device)
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.x2 = h(R, 2 * L.Fp)
        self.x_hat = self.x2 if planes_in else f32(R, L.Fp)
        self.h1 = h(R, 2 * L.Hp)
        self.h1_bits = torch.zeros((R, L.Hp // 8), dtype=torch.uint8, device=device)      # leaky-relu' as one bit per element
        self.z, self.e = f32(R, L.Dp), f32(R, L.Dp)
        self.W1T, self.W2T, self.W2 = h(L.Hp, 2 * L.Fp), h(L.Dp, 2 * L.Hp), h(L.Hp, 2 * L.Dp)
        self.slab_steps = None if backward else 120    # (engine_x3: a forward-only workspace pins the narrow layer's K-slabs)
        nb = max(ops.gemm_f16x2_workspace(False, R, L.Dp, L.Hp), 16)
        if backward:
            self.dz1 = h(R, 2 * L.Hp)
            self.de, self.dz2 = f32(R, L.Dp), f32(R, L.Dp)
            self.dz2_2 = h(R, 2 * L.Dp)
            nb = max(nb, ops.gemm_f16x2_workspace(True, L.Fp, L.Hp, R), ops.gemm_f16x2_workspace(True, L.Hp, L.Dp, R))
        self.gemm_ws = torch.empty(nb // 4, dtype=torch.float32, device=device)
        self.scales = PlaneScales(check_every)
        self._obs = torch.zeros(8, dtype=torch.float32, device=device)
        self.tail_done = False
        self.dz2_planes_done = False

    def _value(self, t, width, scale):
        return (t[:, :width].float() + t[:, width:2 * width].float()) / scale

    def x_hat_f32(self):
        """the gathered, l2-normalised rows as one fp32 tensor [R, Fp] (tests, debugging)"""
        return self.x_hat if self.x_hat.dtype == torch.float32 else self._value(self.x_hat, self.layout.Fp, self.scales.x)

    def h1_f32(self):
        return self._value(self.h1, self.layout.Hp, self.scales.h1)

    def dz1_f32(self):
        return self._value(self.dz1, self.layout.Hp, self.scales.dz1)


def refresh_weights(p, ws):
    """plane copies of the fp32 master weights in the orientations the GEMMs read, at the current weight scales"""
    L, s = p.layout, ws.scales
    ops.split_f32_f16x2(p.W1, ws.W1T, L.Fp, s.w1, transpose=True)        # [Hp][2 Fp]
    ops.split_f32_f16x2(p.W2, ws.W2T, L.Hp, s.w2, transpose=True)        # [Dp][2 Hp]
    ops.split_f32_f16x2(p.W2, ws.W2, L.Dp, s.w2)                         # [Hp][2 Dp]
    if getattr(ws, "W1n", None) is not None:                              # trainable catalogue: dx_hat = dz1 . W1^T reads W1 as it is
        ops.split_f32_f16x2(p.W1, ws.W1n, L.Hp, s.w1)                     # [Fp][2 Hp]


def observe_weights(p, ws):
    """Before a forward pass on a check step: the weights' maxima and the bound of the hidden layer, one device-to-host
    copy; moves the scales that left their window and re-splits the weights if theirs did.  Returns True if any moved."""
    s, o = ws.scales, ws._obs
    o[0] = p.W1.abs().amax()
    o[1] = p.W2.abs().amax()
    o[2] = torch.linalg.vector_norm(p.W1, dim=0).amax() + p.b1.abs().amax()       # |x_hat . W1[:, j] + b1[j]| <= |W1[:, j]| + |b1[j]|
    o[3] = torch.linalg.vector_norm(p.W2, dim=1).amax()                            # (for dz1's bound, kept for observe_gradients)
    hst = [float(v) for v in o[:4].cpu()]
    s.last.update(w1=hst[0], w2=hst[1], h1_bound=hst[2], w2_row=hst[3])
    moved_w = s._move("w1", hst[0], s.TOP_MAX)
    moved_w = s._move("w2", hst[1], s.TOP_MAX) or moved_w
    moved = s._move("h1", hst[2], s.TOP_BOUND) or moved_w
    if moved_w or not s.calibrated:
        refresh_weights(p, ws)
    return moved


def observe_gradients(p, ws):
    """After the loss tail on a check step (ws.dz2 holds the output layer's pre-activation gradient in fp32): dz2's maximum
    and the bound of the hidden layer's gradient.  If dz2's scale moves, the planes the tail wrote are stale: the backward
    pass then splits dz2 again (dz2_planes_done = False)."""
    s, o = ws.scales, ws._obs
    if not ws.tail_done:                                   # a loss path that left de, not dz2: finish the tail here
        ops.l2norm_bwd(ws.z, ws.de, p.layout.Dp, ws.dz2, lrelu_alpha=ops.LRELU_ALPHA)
        ws.tail_done, ws.dz2_planes_done = True, False
    o[4] = ws.dz2.abs().amax()
    o[5] = torch.linalg.vector_norm(ws.dz2, dim=1).amax()
    # the watch on the rule itself: the largest magnitude in the hi planes of this step's h1 and of the LAST step's dz1 -- a value
    # at 65504 means a tensor outgrew its scale between two checks and was clamped (counted and logged; the bounds below re-place
    # the scales either way)
    lo_h, hi_h = torch.aminmax(ws.h1[:, :p.layout.Hp])
    lo_g, hi_g = torch.aminmax(ws.dz1[:, :p.layout.Hp])
    o[6] = torch.maximum(hi_h.float(), -lo_h.float())
    o[7] = torch.maximum(hi_g.float(), -lo_g.float())
    hst = [float(v) for v in o[4:8].cpu()]
    s.last.update(dz2=hst[0], dz2_row=hst[1], h1_hi_plane=hst[2], dz1_hi_plane=hst[3])
    if s.calibrated and max(hst[2], hst[3]) >= 65504.0:
        s.saturated += 1
        import logging
        logging.getLogger("cdml.f16x2").warning("a plane tensor reached fp16's largest value (|h1| hi plane %.0f, |dz1| hi plane %.0f): it "
                                                "outgrew its scale between two checks and was clamped; check more often "
                                                "(PlaneScales.check_every)", hst[2], hst[3])
    if s._move("dz2", hst[0], s.TOP_MAX):
        ws.dz2_planes_done = False
    s._move("dz1", hst[1] * s.last.get("w2_row", 1.0), s.TOP_BOUND)      # |dz2[r] . W2[j]| <= |dz2[r]| |W2[j]|
    if hst[0] > 0.0:                                       # (a batch without one active triplet says nothing about the gradients'
        s.calibrated = True                                #  size: the next step calibrates again)


def tower_forward(p, ws, normalize=True):
    """x_hat (planes, or fp32 -> planes) -> h1 (planes + sign bits) -> z (fp32) -> e (fp32).  models.py:59-61."""
    L, R, s = p.layout, ws.R, ws.scales
    if ws.x_hat.dtype == torch.float32:
        ops.split_f32_f16x2(ws.x_hat, ws.x2, L.Fp, s.x)
    else:
        ws.x2 = ws.x_hat
    ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_X3_BITS, ws.x2, L.Fp, ws.W1T, L.Fp, ws.h1, R, L.Hp, L.Fp, 1.0 / (s.x * s.w1),
                      c_scale=s.h1, plane_c=L.Hp, bias=p.b1, aux=ws.h1_bits)
    ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_F32, ws.h1, L.Hp, ws.W2T, L.Hp, ws.z, R, L.Dp, L.Hp, 1.0 / (s.h1 * s.w2), bias=p.b2,
                      workspace=ws.gemm_ws, slab_steps=ws.slab_steps)
    ws.tail_done = False
    ws.dz2_planes_done = False
    if normalize:
        ops.l2norm_fwd(ws.z, L.Dp, ws.e)
    return ws.e


def tower_backward(p, ws, after_w1=None, w1_chunks=1, after_w1_chunk=None):
    """ws.dz2 (from the fused tail) or ws.de -> p.grad (fp32).  train.py:141; no dX.  ``after_w1`` / ``w1_chunks`` /
    ``after_w1_chunk``: the data-parallel hooks of engine_x3.tower_backward (the first layer's weight gradient first, whole or
    in row blocks of W1, a hook after each -- flat-gradient ranges, the last one ends after db1)."""
    L, R, s = p.layout, ws.R, ws.scales
    if not ws.tail_done:
        ops.l2norm_bwd(ws.z, ws.de, L.Dp, ws.dz2, lrelu_alpha=ops.LRELU_ALPHA)
    if not (ws.tail_done and ws.dz2_planes_done):
        ops.split_f32_f16x2(ws.dz2, ws.dz2_2, L.Dp, s.dz2)
    w2 = lambda: ops.gemm_f16x2_tn(ws.h1, L.Hp, ws.dz2_2, L.Dp, p.gW2, L.Hp, L.Dp, R, 1.0 / (s.h1 * s.dz2), workspace=ws.gemm_ws,
                                   colsum=p.gb2, colsum_scale=1.0 / s.dz2)
    single = after_w1 is None and after_w1_chunk is None
    if single:
        w2()
    # dz1 = (dz2 . W2^T) * lrelu'(h1), written as planes; the sign from FC1's bitmask
    ops.gemm_f16x2_nt(ops.BE_MASKBITS_X3, ws.dz2_2, L.Dp, ws.W2, L.Dp, ws.dz1, R, L.Hp, L.Dp, 1.0 / (s.dz2 * s.w2), c_scale=s.dz1,
                      plane_c=L.Hp, aux=ws.h1_bits)

    def dw1(lo, hi, db):      # columns lo .. hi of both planes of x_hat: the same plane stride, the base moved by lo
        ops.gemm_f16x2_tn(ws.x2[:, lo:], L.Fp, ws.dz1, L.Hp, p.gW1[lo:hi], hi - lo, L.Hp, R, 1.0 / (s.x * s.dz1), workspace=ws.gemm_ws,
                          colsum=db, colsum_scale=1.0 / s.dz1)
    rows = L.Fp // w1_chunks if w1_chunks > 1 else 0
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
