"""Tensor-level wrappers over the C ABI: pointer/stride plumbing only.

Each function hands raw device pointers, sizes and the current torch stream to
libcdml_hip.so.  Nothing here computes; a CPU tensor is an error.
"""
import ctypes as C

import torch

from ._lib import call, load_library

LRELU_ALPHA = 0.2     # tf.nn.leaky_relu default (reference models.py:21)


def _stream():
    # the caller selects the device (TrainStep asserts it); the stream is that device's current one
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=None):
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise ValueError("cdml ops need device tensors (there is no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"expected {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _mat(t, dtype=torch.float32):
    """(pointer, leading dimension) of a 2-D row-major view."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("expected a 2-D tensor with unit inner stride")
    return _p(t, dtype), t.stride(0)


def version():
    return load_library().cdml_version()


# ------------------------------------------------------------------ data ------
def fill_uniform_table(table, row0, feature_size, seed):
    ptr, ld = _mat(table)
    call("cdml_fill_uniform_table", ptr, row0, table.shape[0], feature_size, ld, seed, _stream())
    return table


def sample_uniform(pairs, n_rows, seed, step, batch, idx_out, slot0=0, batch_global=None,
                   step_dev=None):
    bg = batch if batch_global is None else batch_global
    call("cdml_sample_uniform", _p(pairs, torch.int32), pairs.shape[0], n_rows, seed,
         0 if step is None else step, _p(step_dev, torch.int64), batch, slot0, bg,
         _p(idx_out, torch.int32), _stream())
    return idx_out


def sample_inbatch(pairs, seed, step, batch, rows_out, shift_out, slot0=0, batch_global=None,
                   step_dev=None):
    bg = batch if batch_global is None else batch_global
    call("cdml_sample_inbatch", _p(pairs, torch.int32), pairs.shape[0], seed,
         0 if step is None else step, _p(step_dev, torch.int64), batch, slot0, bg,
         _p(rows_out, torch.int32), _p(shift_out, torch.int32), _stream())
    return rows_out, shift_out


def step_advance(step_dev):
    call("cdml_step_advance", _p(step_dev, torch.int64), _stream())


def gather_rows(table, row0, idx, feature_size, x_out, normalize=True, inv_norm_out=None,
                oob_flag=None, nan_missing=False):
    """nan_missing: an idx of -1 is a request that got no row (exchange overflow) -> that output row
    becomes NaN; otherwise -1 marks a padding slot and the output row is left untouched."""
    tp, tld = _mat(table)
    xp, xld = _mat(x_out)
    call("cdml_gather_rows", tp, row0, table.shape[0], tld, _p(idx, torch.int32), idx.numel(),
         feature_size, (1 if normalize else 0) | (2 if nan_missing else 0), xp, xld, _p(inv_norm_out),
         _p(oob_flag, torch.int32), _stream())
    return x_out


def gather_rows_x3(src, idx, feature_size, x_out_planes, nan_missing=False, oob_flag=None):
    """x_out_planes[r] = the three bf16 planes of the fp32 row src[idx[r]] as stored (the row exchange's last step on the
    split-fp32 path: request order and operand form in one pass)."""
    sp, sld = _mat(src)
    xp, xld = _mat16(x_out_planes)
    call("cdml_gather_rows_x3", sp, src.shape[0], sld, _p(idx, torch.int32), idx.numel(), feature_size, 2 if nan_missing else 0,
         xp, xld, _p(oob_flag, torch.int32), _stream())
    return x_out_planes


def sample_gather(mode, pairs, seed, step, batch, table, feature_size, idx_out, x_out,
                  shift_out=None, slot0=0, batch_global=None, step_dev=None, n_steps=1, oob_flag=None, x_ki=None):
    """n_steps > 1: x_out is [n_steps, rows, stride], idx_out [n_steps, rows], shift_out [n_steps]
    (steps step, step+1, ... in one launch).  oob_flag (int32[1]): bit 0 set when a pair id lies
    outside the catalogue (the reference's IndexError, inputs.py:158).  x_ki (three-plane output only): a contiguous
    bf16 buffer [n_steps, 3 * rows * plane] that also receives every step's rows k8-interleaved."""
    bg = batch if batch_global is None else batch_global
    f16 = table.dtype == torch.float16              # fp16 catalogue -> bf16 rows (config 4)
    x3 = not f16 and x_out.dtype == torch.bfloat16   # fp32 catalogue -> rows as three bf16 planes (precision f32x3)
    h2 = not f16 and x_out.dtype == torch.float16    # fp32 catalogue -> rows as two fp16 planes of x_hat * 2^14 (precision f16x2)
    mat = _mat16 if f16 else _mat
    omat = _mat16 if (f16 or x3 or h2) else _mat
    tp, tld = mat(table)
    if n_steps > 1:
        if x_out.dim() != 3 or idx_out.dim() != 2 or x_out.shape[0] != n_steps or idx_out.shape[0] != n_steps:
            raise ValueError("n_steps > 1 needs x_out [n_steps, rows, stride] and idx_out [n_steps, rows]")
        xp, xld = omat(x_out[0])
        xss, iss = x_out.stride(0), idx_out.stride(0)
    else:
        xp, xld = omat(x_out)
        xss = iss = 0
    if x_ki is not None:
        if not x3 or x_ki.dtype != torch.bfloat16 or not x_ki.is_contiguous():
            raise ValueError("x_ki goes with the three-plane output and must be a contiguous bf16 buffer")
        kss = x_ki.stride(0) if (n_steps > 1 and x_ki.dim() > 1) else 0
        call("cdml_sample_gather_x3k", mode, _p(pairs, torch.int32), pairs.shape[0], seed,
             0 if step is None else step, _p(step_dev, torch.int64), batch, slot0, bg, tp,
             table.shape[0], tld, feature_size, _p(idx_out, torch.int32), _p(shift_out, torch.int32),
             xp, xld, n_steps, xss, iss, _p(oob_flag, torch.int32), C.c_void_p(x_ki.data_ptr()), kss, _stream())
        return x_out
    call("cdml_sample_gather_f16" if f16 else "cdml_sample_gather_x3" if x3 else "cdml_sample_gather_h2" if h2 else "cdml_sample_gather",
         mode, _p(pairs, torch.int32), pairs.shape[0], seed,
         0 if step is None else step, _p(step_dev, torch.int64), batch, slot0, bg, tp,
         table.shape[0], tld, feature_size, _p(idx_out, torch.int32), _p(shift_out, torch.int32),
         xp, xld, n_steps, xss, iss, _p(oob_flag, torch.int32), _stream())
    return x_out


def route_rows(ids, rows_per_shard, world, capacity, send_ids, slot_out, overflow_flag):
    call("cdml_route_rows", _p(ids, torch.int32), ids.numel(), rows_per_shard, world, capacity,
         _p(send_ids, torch.int32), _p(slot_out, torch.int32), _p(overflow_flag, torch.int32), _stream())


def scatter_rows(src, slot, dst, width):
    sp, sld = _mat(src)
    dp, dld = _mat(dst)
    call("cdml_scatter_rows", sp, sld, _p(slot, torch.int32), slot.numel(), width, dp, dld, _stream())
    return dst


# ----------------------------------------------------------------- tower ------
def l2norm_fwd(x, n_cols, y, inv_out=None):
    xp, xld = _mat(x)
    yp, yld = _mat(y)
    call("cdml_l2norm_fwd", xp, xld, x.shape[0], n_cols, yp, yld, _p(inv_out), _stream())
    return y


def l2norm_bwd(z, g, n_cols, dz, lrelu_alpha=-1.0):
    zp, zld = _mat(z)
    gp, gld = _mat(g)
    dp, dld = _mat(dz)
    call("cdml_l2norm_bwd", zp, zld, gp, gld, z.shape[0], n_cols, lrelu_alpha, dp, dld, _stream())
    return dz


def fc_lrelu_fwd(x, W, b, y, M, K, N, alpha=LRELU_ALPHA):
    xp, xld = _mat(x)
    wp, wld = _mat(W)
    yp, yld = _mat(y)
    call("cdml_fc_lrelu_fwd", xp, xld, wp, wld, _p(b, torch.float32), alpha, M, K, N, yp, yld,
         _stream())
    return y


def fc_bwd_data(dy, W, x_post, dx, M, K, N, alpha=LRELU_ALPHA):
    dyp, dyld = _mat(dy)
    wp, wld = _mat(W)
    dxp, dxld = _mat(dx)
    if x_post is None:
        xpp, xpld = C.c_void_p(0), 0
    else:
        xpp, xpld = _mat(x_post)
    call("cdml_fc_bwd_data", dyp, dyld, wp, wld, xpp, xpld, alpha, M, K, N, dxp, dxld, _stream())
    return dx


def fc_bwd_weight_workspace(M, K, N):
    return int(load_library().cdml_fc_bwd_weight_workspace(M, K, N))


def fc_bwd_weight(x, dy, dW, db, workspace, M, K, N):
    xp, xld = _mat(x)
    dyp, dyld = _mat(dy)
    wp, wld = _mat(dW)
    call("cdml_fc_bwd_weight", xp, xld, dyp, dyld, M, K, N, wp, wld, _p(db, torch.float32),
         _p(workspace), workspace.numel() * workspace.element_size(), _stream())
    return dW, db


def fc_bwd_weight2_workspace(M, K1, N1, K2, N2):
    """0 = shapes the stream-K launch does not take (use fc_bwd_weight per layer)."""
    return int(load_library().cdml_fc_bwd_weight2_workspace(M, K1, N1, K2, N2))


def fc_bwd_weight2(x1, dy1, dW1, db1, K1, N1, x2, dy2, dW2, db2, K2, N2, M, workspace):
    """Both weight gradients (and bias gradients) of the two-layer tower in one stream-K launch."""
    a1, lda1 = _mat(x1)
    b1, ldb1 = _mat(dy1)
    c1, ldc1 = _mat(dW1)
    a2, lda2 = _mat(x2)
    b2, ldb2 = _mat(dy2)
    c2, ldc2 = _mat(dW2)
    call("cdml_fc_bwd_weight2", a1, lda1, b1, ldb1, K1, N1, c1, ldc1, _p(db1, torch.float32), a2, lda2, b2, ldb2,
         K2, N2, c2, ldc2, _p(db2, torch.float32), M, _p(workspace), workspace.numel() * workspace.element_size(),
         _stream())


# ------------------------------------------------------------------ loss ------
def triplet_hinge(e, B, D, margin, pos, neg, hinge, stats=None, de=None):
    ep, eld = _mat(e)
    dep, deld = (C.c_void_p(0), 0) if de is None else _mat(de)
    call("cdml_triplet_hinge", ep, eld, B, D, margin, _p(pos), _p(neg), _p(hinge), _p(stats), dep,
         deld, _stream())


def triplet_hinge_inbatch(e, rows, shift, B, D, margin, pos, neg, hinge, valid=None, stats=None,
                          de=None):
    ep, eld = _mat(e)
    dep, deld = (C.c_void_p(0), 0) if de is None else _mat(de)
    call("cdml_triplet_hinge_inbatch", ep, eld, _p(rows, torch.int32), _p(shift, torch.int32), B, D,
         margin, _p(pos), _p(neg), _p(hinge), _p(valid, torch.uint8), _p(stats), dep, deld, _stream())


TICKET_WORDS = 128    # CDML_TICKET_WORDS (include/cdml.h)


def new_tickets(device):
    """Zeroed ticket words for cdml_adam_step's advance_step (its last block advances the counter)."""
    return torch.zeros(TICKET_WORDS, dtype=torch.int32, device=device)


def vnet_tail_workspace_floats(B, D):
    return int(load_library().cdml_vnet_tail_workspace(B, D)) // 4


def vnet_tail(mode, z, rows, shift, B, D, margin, e, pos, neg, hinge, dz2, valid=None, stats=None,
              dz2_bf16=None, var_ws=None, alpha=LRELU_ALPHA, plane_bf=0, h2_scale=0.0):
    """l2norm -> hinge loss -> its gradient -> l2norm backward -> lrelu' in one launch
    (mode 0: rows a,p,n per triplet; 1: in-batch negatives).  h2_scale > 0: dz2_bf16 receives the two fp16 planes of
    dz2 * h2_scale (precision f16x2), ``plane_bf`` apart."""
    zp, zld = _mat(z)
    ep, eld = _mat(e)
    dp, dld = _mat(dz2)
    bp, bld = (C.c_void_p(0), 0) if dz2_bf16 is None else _mat16(dz2_bf16)
    if h2_scale:
        call("cdml_vnet_tail_h2", mode, zp, zld, _p(rows, torch.int32), _p(shift, torch.int32), B, D, margin, alpha,
             ep, eld, _p(pos), _p(neg), _p(hinge), _p(valid, torch.uint8), dp, dld, bp, bld, plane_bf, float(h2_scale), _p(stats),
             _p(var_ws), _stream())
        return
    if plane_bf:                                     # dz2 also as its three bf16 planes (precision f32x3)
        call("cdml_vnet_tail_planes", mode, zp, zld, _p(rows, torch.int32), _p(shift, torch.int32), B, D, margin, alpha,
             ep, eld, _p(pos), _p(neg), _p(hinge), _p(valid, torch.uint8), dp, dld, bp, bld, plane_bf, _p(stats),
             _p(var_ws), _stream())
        return
    call("cdml_vnet_tail", mode, zp, zld, _p(rows, torch.int32), _p(shift, torch.int32), B, D, margin, alpha,
         ep, eld, _p(pos), _p(neg), _p(hinge), _p(valid, torch.uint8), dp, dld, bp, bld, _p(stats), _p(var_ws),
         _stream())


def semihard_select(S, e, rows, B, D, sqn_scratch, neg_row_out):
    sp, sld = _mat(S)
    ep, eld = _mat(e)
    call("cdml_semihard_select", sp, sld, ep, eld, _p(rows, torch.int32), B, D, _p(sqn_scratch),
         _p(neg_row_out, torch.int32), _stream())
    return neg_row_out


def semihard_mine_x3_workspace(B):
    return int(load_library().cdml_semihard_mine_x3_workspace(B))


def semihard_mine_x3(e, rows, B, D, e_planes, plane, sqn, dp, workspace, neg_row_out, z=None, h2_scale=0.0):
    """cdml_semihard_select's result without the score matrix: the B x 2B product on the plane kernels, the selection
    as its epilogue (csrc/gemm_bf16x3.hip).  e_planes bf16 [2B, >= 3 plane], sqn f32[2B], dp f32[B], workspace f32.
    ``z`` given: the un-normalised output rows -- the prep launch normalises them and WRITES ``e`` (cdml_semihard_mine_x3_z).
    ``h2_scale`` > 0: the score product on two fp16 planes of e * h2_scale (e_planes fp16 [2B, >= 2 plane]; cdml_semihard_mine_h2)."""
    ep, eld = _mat(e)
    pp, pld = _mat16(e_planes)
    if h2_scale:
        zp, zld = (C.c_void_p(0), 0) if z is None else _mat(z)
        call("cdml_semihard_mine_h2", zp, zld, ep, eld, _p(rows, torch.int32), B, D, pp, pld, plane, float(h2_scale), _p(sqn), _p(dp),
             _p(workspace), workspace.numel() * workspace.element_size(), _p(neg_row_out, torch.int32), _stream())
        return neg_row_out
    if z is not None:
        zp, zld = _mat(z)
        call("cdml_semihard_mine_x3_z", zp, zld, ep, eld, _p(rows, torch.int32), B, D, pp, pld, plane, _p(sqn), _p(dp),
             _p(workspace), workspace.numel() * workspace.element_size(), _p(neg_row_out, torch.int32), _stream())
        return neg_row_out
    call("cdml_semihard_mine_x3", ep, eld, _p(rows, torch.int32), B, D, pp, pld, plane, _p(sqn), _p(dp),
         _p(workspace), workspace.numel() * workspace.element_size(), _p(neg_row_out, torch.int32), _stream())
    return neg_row_out


def triplet_hinge_indexed(e, neg_row, B, D, margin, pos, neg, hinge, scale_scratch, stats=None, de=None, z=None, dz2=None,
                          dz2_bf16=None, plane_bf=0, lrelu_alpha=LRELU_ALPHA):
    """Hinge loss + gradient over (row 2i, row 2i+1, row neg_row[i]).  ``z`` and ``dz2`` given: the finished row gradients
    also go through l2norm_bwd (+ leaky-relu') into dz2 -- and into its bf16 copy / three planes (``dz2_bf16``,
    ``plane_bf``) -- inside the same launch (cdml_triplet_hinge_indexed_tail)."""
    ep, eld = _mat(e)
    dep, deld = (C.c_void_p(0), 0) if de is None else _mat(de)
    if z is None:
        call("cdml_triplet_hinge_indexed", ep, eld, _p(neg_row, torch.int32), B, D, margin, _p(pos), _p(neg),
             _p(hinge), _p(stats), _p(scale_scratch), dep, deld, _stream())
        return
    zp, zld = _mat(z)
    dzp, dzld = _mat(dz2)
    bp, bld = (C.c_void_p(0), 0) if dz2_bf16 is None else _mat16(dz2_bf16)
    call("cdml_triplet_hinge_indexed_tail", ep, eld, _p(neg_row, torch.int32), B, D, margin, _p(pos), _p(neg),
         _p(hinge), _p(stats), _p(scale_scratch), dep, deld, zp, zld, lrelu_alpha, dzp, dzld, bp, bld, plane_bf, _stream())


def pair_dist(e, pairs, D, sqdist, dot, means=None):
    ep, eld = _mat(e)
    call("cdml_pair_dist", ep, eld, e.shape[0], _p(pairs, torch.int32), pairs.shape[0], D, _p(sqdist),
         _p(dot), _p(means), _stream())


# ------------------------------------------------- co-watch graph (N3) --------
def _cowatch_ws(P, device):
    nbytes = int(load_library().cdml_cowatch_workspace(P))
    return torch.empty(nbytes + 256, dtype=torch.uint8, device=device)   # torch allocations are 256-B aligned


def cowatch_graph(pairs):
    P = pairs.shape[0]
    dev = pairs.device
    ws = _cowatch_ws(P, dev)
    edges = torch.empty((P, 2), dtype=torch.int32, device=dev)
    counts = torch.empty(P, dtype=torch.int32, device=dev)
    n_edges = torch.zeros(1, dtype=torch.int64, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    call("cdml_cowatch_graph", _p(pairs, torch.int32), P, _p(edges), _p(counts), _p(n_edges), _p(flag), _p(ws),
         ws.numel(), _stream())
    return edges, counts, n_edges, flag


def cowatch_select(pairs, threshold, unique):
    P = pairs.shape[0]
    dev = pairs.device
    ws = _cowatch_ws(P, dev)
    out = torch.empty((P, 2), dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    call("cdml_cowatch_select", _p(pairs, torch.int32), P, threshold, 1 if unique else 0, _p(out), _p(n), _p(flag),
         _p(ws), ws.numel(), _stream())
    return out, n, flag


# ------------------------------------------------- exact kNN export (N4) ------
def knn_list_capacity():
    return int(load_library().cdml_knn_list_capacity())


def row_sqnorm(x, D, out):
    xp, xld = _mat(x)
    call("cdml_row_sqnorm", xp, xld, x.shape[0], D, _p(out), _stream())
    return out


def knn_merge(scores, nq, nb, col0, n_valid, q_sq, b_sq, k, best_d, best_i, first):
    sp, sld = _mat(scores)
    call("cdml_knn_merge", sp, sld, nq, nb, col0, n_valid, _p(q_sq), _p(b_sq), k, _p(best_d),
         _p(best_i, torch.int32), int(bool(first)), _stream())


def knn_filter_x3(Q3, plane_q, B3, plane_b, nq, n_cols, D, q_sq, b_sq, tau, col0, n_valid, cnt, cand, cap):
    """The query x catalogue-block inner products on the plane kernels with the threshold filter as their epilogue: every
    element with d <= tau[query] is appended to the query's candidate list (cand int32 [nq, cap, 2]: float bits of d, id)."""
    qp, qld = _mat16(Q3)
    bp, bld = _mat16(B3)
    if cand.dtype != torch.int32 or not cand.is_contiguous() or cand.numel() < nq * cap * 2:
        raise ValueError("cand must be a contiguous int32 buffer of nq * cap * 2 words")
    call("cdml_knn_filter_x3", qp, qld, plane_q, bp, bld, plane_b, nq, n_cols, D, _p(q_sq), _p(b_sq), _p(tau), col0, n_valid,
         _p(cnt, torch.int32), _p(cand, torch.int32), cap, _stream())


def knn_filter_h2(Q2, plane_q, B2, plane_b, nq, n_cols, D, out_scale, q_sq, b_sq, tau, col0, n_valid, cnt, cand, cap):
    """knn_filter_x3 on two fp16 planes per row (split_f32_f16x2 at scales sq / sb; out_scale = 1 / (sq sb))."""
    qp, qld = _mat16(Q2)
    bp, bld = _mat16(B2)
    if cand.dtype != torch.int32 or not cand.is_contiguous() or cand.numel() < nq * cap * 2:
        raise ValueError("cand must be a contiguous int32 buffer of nq * cap * 2 words")
    call("cdml_knn_filter_h2", qp, qld, plane_q, bp, bld, plane_b, nq, n_cols, D, float(out_scale), _p(q_sq), _p(b_sq), _p(tau), col0,
         n_valid, _p(cnt, torch.int32), _p(cand, torch.int32), cap, _stream())


def knn_merge_list(cand, cnt, cap, nq, k, best_d, best_i, overflow):
    call("cdml_knn_merge_list", _p(cand, torch.int32), _p(cnt, torch.int32), cap, nq, k, _p(best_d), _p(best_i, torch.int32),
         _p(overflow, torch.int32), _stream())


# ------------------------------------------------- fusion towers (N4) ---------
EW_MUL, EW_MUL_RES, EW_ADD = 0, 1, 2


def ew_combine(mode, a, b, out, M, N):
    ap, ald = _mat(a)
    bp, bld = _mat(b)
    op, old = _mat(out)
    call("cdml_ew_combine", mode, ap, ald, bp, bld, M, N, op, old, _stream())
    return out


def ew_fusion_bwd(residual, g, a, b, da, db, M, N, alpha=LRELU_ALPHA):
    gp, gld = _mat(g)
    ap, ald = _mat(a)
    bp, bld = _mat(b)
    dap, dald = _mat(da)
    dbp, dbld = _mat(db)
    call("cdml_ew_fusion_bwd", 1 if residual else 0, gp, gld, ap, ald, bp, bld, M, N, alpha, dap, dald,
         dbp, dbld, _stream())


def lrelu_bwd(g, y, out, M, N, alpha=LRELU_ALPHA):
    gp, gld = _mat(g)
    yp, yld = _mat(y)
    op, old = _mat(out)
    call("cdml_lrelu_bwd", gp, gld, yp, yld, M, N, alpha, op, old, _stream())
    return out


# ------------------------------------------- reduced precision (config 4) -----
BE_BIAS_LRELU_BF16, BE_BIAS_LRELU_F32, BE_MASK_BF16, BE_F32 = 0, 1, 2, 3
BE_BIAS_LRELU_BF16_BITS, BE_MASKBITS_BF16 = 4, 5      # 0 + sign bitmask out (aux) / 2 reading that bitmask


def gemm_bf16_epilogue_supported(epilogue, M, N, K, lda, ldb, ldc, ldaux):
    return bool(load_library().cdml_gemm_bf16_epilogue_supported(epilogue, M, N, K, lda, ldb, ldc, ldaux))


def _mat16(t):
    if t.dim() != 2 or t.stride(1) != 1 or t.element_size() != 2:
        raise ValueError("expected a 2-D 16-bit tensor with unit inner stride")
    return _p(t), t.stride(0)


def gemm_bf16_workspace(M, N, K):
    return int(load_library().cdml_gemm_bf16_workspace(M, N, K))


def gemm_bf16_nt(epilogue, A, B, C, M, N, K, bias=None, aux=None, alpha=LRELU_ALPHA, workspace=None):
    ap, ald = _mat16(A)
    bp, bld = _mat16(B)
    if C.dim() != 2 or C.stride(1) != 1:
        raise ValueError("C must be 2-D with unit inner stride")
    xld = aux.stride(0) if aux is not None else 0
    if aux is not None and (aux.dtype == torch.uint8) != (epilogue in (BE_BIAS_LRELU_BF16_BITS, BE_MASKBITS_BF16)):
        raise ValueError("epilogues 4 / 5 take a uint8 bitmask as aux, epilogue 2 bf16 values")
    call("cdml_gemm_bf16_nt", epilogue, ap, ald, bp, bld, M, N, K, _p(C), C.stride(0), _p(bias),
         _p(aux), xld, alpha, _p(workspace), 0 if workspace is None else workspace.numel() * workspace.element_size(),
         _stream())
    return C


def gemm_bf16_tn_supported(M, N, K, lda, ldb):
    return bool(load_library().cdml_gemm_bf16_tn_supported(M, N, K, lda, ldb))


def gemm_bf16_tn_workspace(M, N, K):
    return int(load_library().cdml_gemm_bf16_tn_workspace(M, N, K))


def gemm_bf16_tn(A, B, C, M, N, K, workspace=None, colsum=None):
    """C[M][N] f32 = sum_k A[k][M] * B[k][N]: the weight gradient from the activations as
    stored; colsum[n] = sum_k B[k][n] (the bias gradient) on request."""
    ap, ald = _mat16(A)
    bp, bld = _mat16(B)
    cp, cld = _mat(C)
    call("cdml_gemm_bf16_tn", ap, ald, bp, bld, M, N, K, cp, cld, _p(colsum, torch.float32), _p(workspace),
         0 if workspace is None else workspace.numel() * workspace.element_size(), _stream())
    return C


# ---- fp32 products on the bf16 MFMA: operands as three bf16 planes (csrc/gemm_bf16x3.hip) ----
BE_BIAS_LRELU_X3, BE_MASK_X3, BE_ROWBIAS_LRELU_X3 = 6, 7, 8
BE_BIAS_LRELU_X3_BITS, BE_MASKBITS_X3 = 9, 10         # 6 + sign bitmask out (aux, uint8 [M][N / 8]) / 7 reading that bitmask
BE_MASKBITS_X3_KI = 12                                # 10 (7 without aux) with the result's planes k8-interleaved: [3][M / 8][ldc][8]


def split_f32_bf16x3(src, dst, plane, transpose=False):
    """dst[r][p*plane + c] = bf16 plane p (hi, mid, lo) of the fp32 src[r][c]; transpose: dst[c][p*plane + r]."""
    sp, sld = _mat(src)
    dp, dld = _mat16(dst)
    call("cdml_split_f32_bf16x3", sp, sld, src.shape[0], src.shape[1], dp, dld, plane, 1 if transpose else 0, _stream())
    return dst


def gemm_bf16x3_workspace(tn, M, N, K, products=6):
    return int(load_library().cdml_gemm_bf16x3_workspace(1 if tn else 0, M, N, K, products))


def gemm_bf16x3_nt(epilogue, A, plane_a, B, plane_b, C, M, N, K, products=6, plane_c=0, bias=None, aux=None,
                   alpha=LRELU_ALPHA, workspace=None, colsum=None, ldc=None, slab_steps=None):
    """C = epilogue(A . B^T) for fp32 operands given as bf16 planes [rows][hi K | mid K | lo K]; colsum[n] = sum_k
    B[n][k] on request.  Epilogue 12: C is the flat k8-interleaved buffer [3][M / 8][ldc][8] (``ldc`` = columns per row
    group, ``plane_c`` = elements per plane).  ``slab_steps`` (the narrow layer, N = 256): pin the K-slab length for this
    call (cdml_x3_slab_steps) instead of the rule by row-tile class."""
    if slab_steps:
        lib = load_library()
        prev = lib.cdml_x3_slab_steps(int(slab_steps))
        try:
            return gemm_bf16x3_nt(epilogue, A, plane_a, B, plane_b, C, M, N, K, products=products, plane_c=plane_c, bias=bias,
                                  aux=aux, alpha=alpha, workspace=workspace, colsum=colsum, ldc=ldc)
        finally:
            lib.cdml_x3_slab_steps(prev)
    ap, ald = _mat16(A)
    bp, bld = _mat16(B)
    if epilogue == BE_MASKBITS_X3_KI:
        if ldc is None or not C.is_contiguous() or C.dtype != torch.bfloat16:
            raise ValueError("epilogue 12 writes a contiguous bf16 buffer and needs ldc (columns per row group)")
    elif C.dim() != 2 or C.stride(1) != 1:
        raise ValueError("C must be 2-D with unit inner stride")
    if aux is not None and (aux.dtype == torch.uint8) != (epilogue in (BE_BIAS_LRELU_X3_BITS, BE_MASKBITS_X3, BE_MASKBITS_X3_KI)):
        raise ValueError("epilogues 9 / 10 / 12 take a uint8 bitmask as aux, epilogue 7 bf16 values")
    call("cdml_gemm_bf16x3_nt", epilogue, ap, ald, plane_a, bp, bld, plane_b, M, N, K, products, _p(C), C.stride(0) if ldc is None else ldc,
         plane_c, _p(bias), _p(aux), aux.stride(0) if aux is not None else 0, alpha, _p(colsum, torch.float32),
         _p(workspace), 0 if workspace is None else workspace.numel() * workspace.element_size(), _stream())
    return C


def interleave8_bf16x3(src, plane_src, rows, cols, dst):
    """row-major planes [rows, >= 2 plane_src + cols] -> k8-interleaved [3, rows / 8, cols, 8] (csrc/gemm_bf16x3.hip)"""
    sp, sld = _mat16(src)
    if dst.dtype != torch.bfloat16 or not dst.is_contiguous() or dst.numel() < 3 * rows * cols:
        raise ValueError("interleave8_bf16x3: dst must be a contiguous bf16 buffer of 3 * rows * cols elements")
    call("cdml_interleave8_bf16x3", sp, sld, plane_src, rows, cols, C.c_void_p(dst.data_ptr()), _stream())
    return dst


def gemm_bf16x3_tnk(A, ma, a_col0, B, nb, b_col0, out, M, N, K, workspace=None, colsum=None):
    """out[M][N] f32 = sum_k A[k][a_col0 + m] B[k][b_col0 + n] on k8-interleaved operands [3, K / 8, ma | nb, 8] (six products)."""
    cp, cld = _mat(out)
    ws, wb = (C.c_void_p(0), 0) if workspace is None else (C.c_void_p(workspace.data_ptr()), workspace.numel() * workspace.element_size())
    call("cdml_gemm_bf16x3_tnk", C.c_void_p(A.data_ptr()), ma, a_col0, C.c_void_p(B.data_ptr()), nb, b_col0, M, N, K, cp, cld,
         _p(colsum, torch.float32), ws, wb, _stream())
    return out


def gemm_bf16x3_tn(A, plane_a, B, plane_b, C, M, N, K, products=6, workspace=None, colsum=None, bias=None,
                   alpha=LRELU_ALPHA):
    """C[M][N] f32 = sum_k A[k][M] B[k][N] for fp32 operands given as bf16 planes [K][hi | mid | lo]
    (bias given: C = lrelu(. + bias))."""
    ap, ald = _mat16(A)
    bp, bld = _mat16(B)
    cp, cld = _mat(C)
    call("cdml_gemm_bf16x3_tn", ap, ald, plane_a, bp, bld, plane_b, M, N, K, products, cp, cld, _p(bias), alpha,
         _p(colsum, torch.float32), _p(workspace),
         0 if workspace is None else workspace.numel() * workspace.element_size(), _stream())
    return C


# ---- two fp16 planes per fp32 operand, three plane products (precision "f16x2"; csrc/gemm_f16x2_256.hip) ----
def split_f32_f16x2(src, dst, plane, scale, transpose=False):
    """dst[r][p*plane + c] = fp16 plane p (hi, lo) of src[r][c] * scale; transpose: dst[c][p*plane + r]."""
    sp, sld = _mat(src)
    dp, dld = _mat16(dst)
    call("cdml_split_f32_f16x2", sp, sld, src.shape[0], src.shape[1], dp, dld, plane, 1 if transpose else 0, float(scale), _stream())
    return dst


def gemm_f16x2_workspace(tn, M, N, K):
    return int(load_library().cdml_gemm_f16x2_workspace(1 if tn else 0, M, N, K))


def gemm_f16x2_nt(epilogue, A, plane_a, B, plane_b, C, M, N, K, out_scale, c_scale=1.0, plane_c=0, bias=None, aux=None,
                  alpha=LRELU_ALPHA, workspace=None, slab_steps=None):
    """C = epilogue(out_scale * A . B^T) for operands given as fp16 planes [rows][hi K | lo K] of (value * its scale);
    plane outputs (epilogues 6 / 7 / 9 / 10) are the fp16 planes of (result * c_scale)."""
    if slab_steps:
        lib = load_library()
        prev = lib.cdml_x3_slab_steps(int(slab_steps))
        try:
            return gemm_f16x2_nt(epilogue, A, plane_a, B, plane_b, C, M, N, K, out_scale, c_scale=c_scale, plane_c=plane_c,
                                 bias=bias, aux=aux, alpha=alpha, workspace=workspace)
        finally:
            lib.cdml_x3_slab_steps(prev)
    ap, ald = _mat16(A)
    bp, bld = _mat16(B)
    if C.dim() != 2 or C.stride(1) != 1:
        raise ValueError("C must be 2-D with unit inner stride")
    if aux is not None and (aux.dtype == torch.uint8) != (epilogue in (BE_BIAS_LRELU_X3_BITS, BE_MASKBITS_X3)):
        raise ValueError("epilogues 9 / 10 take a uint8 bitmask as aux, epilogue 7 16-bit values")
    call("cdml_gemm_f16x2_nt", epilogue, ap, ald, plane_a, bp, bld, plane_b, M, N, K, _p(C), C.stride(0), plane_c, _p(bias),
         _p(aux), aux.stride(0) if aux is not None else 0, alpha, float(out_scale), float(c_scale), _p(workspace),
         0 if workspace is None else workspace.numel() * workspace.element_size(), _stream())
    return C


def gemm_f16x2_tn(A, plane_a, B, plane_b, C, M, N, K, out_scale, workspace=None, colsum=None, colsum_scale=1.0):
    """C[M][N] f32 = out_scale * sum_k A[k][M] B[k][N] on fp16 planes [K][hi | lo]; colsum[n] = colsum_scale * sum_k B[k][n]."""
    ap, ald = _mat16(A)
    bp, bld = _mat16(B)
    cp, cld = _mat(C)
    call("cdml_gemm_f16x2_tn", ap, ald, plane_a, bp, bld, plane_b, M, N, K, cp, cld, float(out_scale), _p(colsum, torch.float32),
         float(colsum_scale), _p(workspace), 0 if workspace is None else workspace.numel() * workspace.element_size(), _stream())
    return C


def gemm_bf16_tn2_workspace(M1, N1, M2, N2, K):
    """0 = shapes the joint launch does not take (use gemm_bf16_tn per product)."""
    return int(load_library().cdml_gemm_bf16_tn2_workspace(M1, N1, M2, N2, K))


def gemm_bf16_tn2(A1, B1, C1, M1, N1, A2, B2, C2, M2, N2, K, workspace, colsum1=None, colsum2=None):
    """C1 = A1^T.B1 and C2 = A2^T.B2 (k-strided operands, one K) in one stream-K launch + fix-up pass."""
    a1, lda1 = _mat16(A1)
    b1, ldb1 = _mat16(B1)
    c1, ldc1 = _mat(C1)
    a2, lda2 = _mat16(A2)
    b2, ldb2 = _mat16(B2)
    c2, ldc2 = _mat(C2)
    call("cdml_gemm_bf16_tn2", a1, lda1, b1, ldb1, M1, N1, c1, ldc1, _p(colsum1, torch.float32), a2, lda2, b2, ldb2,
         M2, N2, c2, ldc2, _p(colsum2, torch.float32), K, _p(workspace), workspace.numel() * workspace.element_size(),
         _stream())


def transpose_to_bf16(src, dst, rows, cols):
    call("cdml_transpose_to_bf16", 1 if src.dtype == torch.float32 else 0, _p(src), src.stride(0), rows, cols,
         _p(dst, torch.bfloat16), dst.stride(0), _stream())
    return dst


def cast_f32_bf16(src, dst, rows, cols):
    call("cdml_cast_f32_bf16", _p(src, torch.float32), src.stride(0), rows, cols, _p(dst, torch.bfloat16),
         dst.stride(0), _stream())
    return dst


def colsum_workspace_floats(rows, cols):
    return int(load_library().cdml_colsum_workspace_floats(rows, cols))


def colsum(src, rows, cols, out, workspace):
    call("cdml_colsum", 1 if src.dtype == torch.bfloat16 else 0, _p(src), src.stride(0), rows, cols,
         _p(out, torch.float32), _p(workspace, torch.float32), _stream())
    return out


def fill_uniform_table_f16(table, row0, feature_size, seed):
    call("cdml_fill_uniform_table_f16", _p(table, torch.float16), row0, table.shape[0], feature_size,
         table.stride(0), seed, _stream())
    return table


def gather_rows_f16(table, row0, idx, feature_size, x_out, oob_flag=None):
    call("cdml_gather_rows_f16", _p(table, torch.float16), row0, table.shape[0], table.stride(0),
         _p(idx, torch.int32), idx.numel(), feature_size, _p(x_out, torch.bfloat16), x_out.stride(0),
         _p(oob_flag, torch.int32), _stream())
    return x_out


# ------------------------------------------------------------- optimizers -----
def adam_step(w, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, lr_dev=None, t_dev=None,
              advance_tickets=None):
    """advance_tickets (ops.new_tickets): also do global_step += 1 on *t_dev in the same launch."""
    call("cdml_adam_step", _p(w), _p(g), _p(m), _p(v), w.numel(), lr, _p(lr_dev), beta1, beta2, eps,
         0 if t is None else t, _p(t_dev, torch.int64), 0 if advance_tickets is None else 1,
         _p(advance_tickets, torch.int32), _stream())


def adam_matrix_bf16(W, g, m, v, lr, t, wt=None, wc=None, beta1=0.9, beta2=0.999, eps=1e-8, lr_dev=None, t_dev=None,
                     bias=None, advance_tickets=None, plane_t=0, plane_c=0, h2_scale=0.0):
    """Adam on the contiguous weight matrix W [K, N] (g, m, v alike) that also writes the bf16 operand
    copies: wt = W^T as bf16 [N, >=K], wc = W as bf16 [K, >=N] (either may be None).  ``bias`` =
    (b, gb, mb, vb): the layer's bias vector updated in the same launch; ``advance_tickets``
    (ops.new_tickets): also global_step += 1 on *t_dev by the last block."""
    if W.dim() != 2 or not W.is_contiguous() or W.dtype != torch.float32:
        raise ValueError("W must be a contiguous fp32 matrix")
    K, N = W.shape
    for name, x in (("g", g), ("m", m), ("v", v)):
        if x.numel() != K * N or not x.is_contiguous() or x.dtype != torch.float32:
            raise ValueError("%s must be contiguous fp32 with W's size" % name)
    tp, tld = (C.c_void_p(0), 0) if wt is None else _mat16(wt)
    cp, cld = (C.c_void_p(0), 0) if wc is None else _mat16(wc)
    planes = bool(plane_t or plane_c)                # the copies as three bf16 planes (precision f32x3)
    np1 = 1 if h2_scale else 2                       # (h2_scale: as two fp16 planes of W * h2_scale, precision f16x2)
    if wt is not None and (wt.shape[0] < N or wt.shape[1] < (np1 * plane_t + K if planes else K)):
        raise ValueError("wt must be at least [N, K] (planes: [N, 2 plane_t + K])")
    if wc is not None and (wc.shape[0] < K or wc.shape[1] < (np1 * plane_c + N if planes else N)):
        raise ValueError("wc must be at least [K, N] (planes: [K, 2 plane_c + N])")
    b = bias if bias is not None else (None, None, None, None)
    if bias is not None and any(x.numel() != b[0].numel() or not x.is_contiguous() for x in b):
        raise ValueError("bias, its gradient and its moments must be contiguous vectors of one size")
    if h2_scale:
        call("cdml_adam_matrix_h2", _p(W), _p(g), _p(m), _p(v), K, N, lr, _p(lr_dev), beta1, beta2, eps,
             0 if t is None else t, _p(t_dev, torch.int64), tp, tld, plane_t, cp, cld, plane_c, float(h2_scale), _p(b[0]), _p(b[1]),
             _p(b[2]), _p(b[3]), 0 if bias is None else b[0].numel(), 0 if advance_tickets is None else 1,
             _p(advance_tickets, torch.int32), _stream())
        return
    if planes:
        call("cdml_adam_matrix_planes", _p(W), _p(g), _p(m), _p(v), K, N, lr, _p(lr_dev), beta1, beta2, eps,
             0 if t is None else t, _p(t_dev, torch.int64), tp, tld, plane_t, cp, cld, plane_c, _p(b[0]), _p(b[1]),
             _p(b[2]), _p(b[3]), 0 if bias is None else b[0].numel(), 0 if advance_tickets is None else 1,
             _p(advance_tickets, torch.int32), _stream())
        return
    call("cdml_adam_matrix_bf16", _p(W), _p(g), _p(m), _p(v), K, N, lr, _p(lr_dev), beta1, beta2, eps,
         0 if t is None else t, _p(t_dev, torch.int64), tp, tld, cp, cld, _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]),
         0 if bias is None else b[0].numel(), 0 if advance_tickets is None else 1, _p(advance_tickets, torch.int32),
         _stream())


def table_adam_rows(table, row0, F, idx, grad_xhat, m_table, v_table, head, nxt, lr, t, beta1=0.9, beta2=0.999,
                    eps=1e-8, lr_dev=None, t_dev=None, grad_scale=1.0):
    """Lazy-Adam update of the catalogue rows a batch touched (see include/cdml.h)."""
    tp, tld = _mat(table)
    gp, gld = _mat(grad_xhat)
    if m_table.shape != table.shape or v_table.shape != table.shape or m_table.stride(0) != tld \
            or v_table.stride(0) != tld:
        raise ValueError("m/v tables must have the table's shape and stride")
    if head.numel() < table.shape[0] or nxt.numel() < idx.numel():
        raise ValueError("head needs one int32 per table row, next one per gathered row")
    call("cdml_table_adam_rows", tp, row0, table.shape[0], tld, F, _p(idx, torch.int32), idx.numel(), gp, gld,
         _p(m_table), _p(v_table), _p(head, torch.int32), _p(nxt, torch.int32), grad_scale, lr, _p(lr_dev), beta1, beta2,
         eps, 0 if t is None else t, _p(t_dev, torch.int64), _stream())


def grad_prepare(g, w, l2_scale, clip_norm, scratch, norms_out=None):
    """In place on one variable: g += l2_scale*w, then tf.clip_by_norm(g, clip_norm)."""
    call("cdml_grad_prepare", _p(g), _p(w), g.numel(), l2_scale, clip_norm, _p(scratch), _p(norms_out), _stream())


def momentum_step(w, g, acc, lr, momentum=0.9, use_nesterov=True, lr_dev=None):
    call("cdml_momentum_step", _p(w), _p(g), _p(acc), w.numel(), lr, _p(lr_dev), momentum,
         1 if use_nesterov else 0, _stream())


def lars_scratch_floats():
    return int(load_library().cdml_lars_scratch_floats())


def lars_step(w, g, acc, lr, scratch, momentum=0.9, weight_decay=1e-4, eeta=1e-3, eps=0.0,
              lr_dev=None):
    call("cdml_lars_step", _p(w), _p(g), _p(acc), w.numel(), lr, _p(lr_dev), momentum, weight_decay,
         eeta, eps, _p(scratch), _stream())


def lars_multi_scratch_floats():
    return int(load_library().cdml_lars_multi_scratch_floats())


def lars_multi(w, g, acc, segments, lr, scratch, momentum=0.9, weight_decay=1e-4, eeta=1e-3, eps=0.0,
               lr_dev=None, norms_out=None, step_dev=None, tickets=None):
    """LARS on every variable of the flat buffer in two launches: ``segments`` = [(offset, numel), ...]
    tiling w contiguously; step_dev (with tickets): also global_step += 1."""
    n = len(segments)
    offs = (C.c_int64 * n)(*[int(o) for o, _ in segments])
    sizes = (C.c_int64 * n)(*[int(m) for _, m in segments])
    call("cdml_lars_multi", _p(w), _p(g), _p(acc), C.cast(offs, C.c_void_p), C.cast(sizes, C.c_void_p), n, lr,
         _p(lr_dev), momentum, weight_decay, eeta, eps, _p(scratch), _p(norms_out), _p(step_dev, torch.int64),
         _p(tickets, torch.int32), _stream())


def _seg_arrays(segments):
    n = len(segments)
    offs = (C.c_int64 * n)(*[int(o) for o, _ in segments])
    sizes = (C.c_int64 * n)(*[int(m) for _, m in segments])
    return n, offs, sizes


def lars_multi_norms(w, g, segments, scratch):
    """Launch 1 of LARS alone: the per-block |w|^2, |g|^2 partials of every segment (then ``lars_matrix`` per matrix)."""
    n, offs, sizes = _seg_arrays(segments)
    call("cdml_lars_multi_norms", _p(w), _p(g), C.cast(offs, C.c_void_p), C.cast(sizes, C.c_void_p), n, _p(scratch), _stream())


def _copies(wt, wc, K, N, plane_t, plane_c, h2_scale=0.0):
    tp, tld = (C.c_void_p(0), 0) if wt is None else _mat16(wt)
    cp, cld = (C.c_void_p(0), 0) if wc is None else _mat16(wc)
    planes = 2 if h2_scale else 3 if (plane_t or plane_c) else 1
    if wt is not None and (wt.shape[0] < N or wt.shape[1] < ((planes - 1) * plane_t + K if planes > 1 else K)):
        raise ValueError("wt must be at least [N, K] (planes: [N, 2 plane_t + K]; fp16 planes: [N, plane_t + K])")
    if wc is not None and (wc.shape[0] < K or wc.shape[1] < ((planes - 1) * plane_c + N if planes > 1 else N)):
        raise ValueError("wc must be at least [K, N] (planes: [K, 2 plane_c + N]; fp16 planes: [K, plane_c + N])")
    return tp, tld, cp, cld, planes


def lars_matrix(w, g, acc, segments, seg_matrix, seg_bias, K, N, lr, scratch, wt=None, wc=None, plane_t=0, plane_c=0,
                momentum=0.9, weight_decay=1e-4, eeta=1e-3, eps=0.0, lr_dev=None, norms_out=None, step_dev=None,
                tickets=None, h2_scale=0.0):
    """LARS on the K x N weight matrix in segment ``seg_matrix`` of the flat buffers (+ the bias vector in segment
    ``seg_bias``, or None), after ``lars_multi_norms`` on the same segments; also writes the operand copies wt = W^T,
    wc = W as bf16 (plane_t / plane_c > 0: as three bf16 planes).  step_dev (with tickets): also global_step += 1."""
    n, offs, sizes = _seg_arrays(segments)
    tp, tld, cp, cld, planes = _copies(wt, wc, K, N, plane_t, plane_c, h2_scale)
    if h2_scale:                                     # the copies as two fp16 planes of W * h2_scale (precision f16x2)
        call("cdml_lars_matrix_h2", _p(w), _p(g), _p(acc), C.cast(offs, C.c_void_p), C.cast(sizes, C.c_void_p), n, seg_matrix,
             -1 if seg_bias is None else seg_bias, K, N, lr, _p(lr_dev), momentum, weight_decay, eeta, eps, _p(scratch),
             _p(norms_out), tp, tld, plane_t, cp, cld, plane_c, float(h2_scale), _p(step_dev, torch.int64),
             _p(tickets, torch.int32), _stream())
        return
    call("cdml_lars_matrix", _p(w), _p(g), _p(acc), C.cast(offs, C.c_void_p), C.cast(sizes, C.c_void_p), n, seg_matrix,
         -1 if seg_bias is None else seg_bias, K, N, lr, _p(lr_dev), momentum, weight_decay, eeta, eps, _p(scratch),
         _p(norms_out), tp, tld, plane_t, cp, cld, plane_c, planes, _p(step_dev, torch.int64), _p(tickets, torch.int32),
         _stream())


def momentum_matrix(W, g, acc, lr, wt=None, wc=None, plane_t=0, plane_c=0, momentum=0.9, use_nesterov=True, lr_dev=None,
                    bias=None, step_dev=None, tickets=None, h2_scale=0.0):
    """ApplyMomentum on the contiguous weight matrix W [K, N] (g, acc alike) + ``bias`` = (b, gb, accb), writing the
    operand copies like ``lars_matrix``."""
    if W.dim() != 2 or not W.is_contiguous() or W.dtype != torch.float32:
        raise ValueError("W must be a contiguous fp32 matrix")
    K, N = W.shape
    tp, tld, cp, cld, planes = _copies(wt, wc, K, N, plane_t, plane_c, h2_scale)
    b = bias if bias is not None else (None, None, None)
    if h2_scale:
        call("cdml_momentum_matrix_h2", _p(W), _p(g), _p(acc), K, N, lr, _p(lr_dev), momentum, 1 if use_nesterov else 0, tp, tld,
             plane_t, cp, cld, plane_c, float(h2_scale), _p(b[0]), _p(b[1]), _p(b[2]), 0 if bias is None else b[0].numel(),
             _p(step_dev, torch.int64), _p(tickets, torch.int32), _stream())
        return
    call("cdml_momentum_matrix", _p(W), _p(g), _p(acc), K, N, lr, _p(lr_dev), momentum, 1 if use_nesterov else 0, tp, tld,
         plane_t, cp, cld, plane_c, planes, _p(b[0]), _p(b[1]), _p(b[2]), 0 if bias is None else b[0].numel(),
         _p(step_dev, torch.int64), _p(tickets, torch.int32), _stream())
