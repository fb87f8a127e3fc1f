"""Exact k-nearest-neighbour export behind the reference's ``calc_knn`` signature
(faiss_knn.py:82-131).

The reference l2-normalises the embeddings, builds a faiss ``IndexHNSWFlat`` and
returns ``D`` (squared L2 distances, ascending) and ``I`` (int64 ids; the query
itself is its own first neighbour -- "51 = 50 neighbours + the query").  HNSW is an
approximation of the exact answer; this module computes the exact one by brute
force on the GPU: blocks of inner products ``Q @ Bᵀ`` from the plane kernels
(the headline path's arithmetic, ``ops.gemm_bf16x3_nt``; or the fp32 MFMA GEMM,
``ops.fc_bwd_data`` without a mask), folded into per-query candidate lists by
``ops.knn_merge`` while the 128-MiB score block is still in the Infinity Cache.  Ties are ordered by id.  ``M``,
``efConstruction`` and ``efSearch`` are accepted for call compatibility and ignored.
"""
import numpy as np
import torch

from . import ops

Q_BLOCK = 4096      # query rows per GEMM; 4096 x 8192 scores = 128 MiB stay in the 256 MiB Infinity Cache
B_BLOCK = 8192
# Round 6 (precision "f32x3"): only the FIRST block of the catalogue goes through score blocks -- it gives every query a k-th
# best distance tau -- and the rest through ONE launch of the plane GEMM whose epilogue appends every element within tau to
# its query's candidate list (cdml_knn_filter_x3): no score matrix.  An element of the rest passes with probability
# ~ k / FIRST_BLOCK for exchangeable rows, so a query collects ~ k * (n - FIRST_BLOCK) / FIRST_BLOCK candidates; the lists
# have four times that many slots, and a list that overflows anyway (rows sorted so that later ones are closer) sends the
# whole search back through the score blocks -- nothing is silently lost.
FIRST_BLOCK = 32768


def _round_up(x, m):
    return (x + m - 1) // m * m


def _device_matrix(a, device, row_multiple, col_multiple=32):
    """[n, D] -> zero-padded fp32 device matrix [n_pad, Dp] (Dp % col_multiple == 0)."""
    t = a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a, dtype=np.float32))
    t = t.to(device=device, dtype=torch.float32)
    n, D = t.shape
    out = torch.zeros((_round_up(n, row_multiple), _round_up(D, col_multiple)), dtype=torch.float32, device=device)
    out[:n, :D] = t
    return out


def _planes(x, Dp):
    out = torch.empty((x.shape[0], 3 * Dp), dtype=torch.bfloat16, device=x.device)
    ops.split_f32_bf16x3(x, out, Dp)
    return out


def _planes_h2(x, Dp):
    """two fp16 planes of x * 2^s, s from the tensor's maximum (a host read: the export is not a step path); (planes, scale)"""
    from .engine_f16x2 import pow2_for
    scale = pow2_for(float(x.abs().amax()), 2.0 ** 13) or 1.0
    out = torch.empty((x.shape[0], 2 * Dp), dtype=torch.float16, device=x.device)
    ops.split_f32_f16x2(x, out, Dp, scale)
    return out, scale


def knn_search(base, queries, k, l2_norm=True, device="cuda:0", q_block=Q_BLOCK, b_block=B_BLOCK, precision="f32x3",
               fused=True, first_block=FIRST_BLOCK, q_chunk=262144, c_chunk=1048576):
    """Device tensors (D [nq,k] fp32 squared L2 ascending, I [nq,k] int64; -1 where
    the catalogue has fewer than k rows).  ``precision``: "f32x3" (default; round 6) = the inner products on the plane
    kernels -- the headline path's arithmetic: every fp32 operand as three exact bf16 planes, six plane products per fp32
    product on the bf16 MFMA (csrc/gemm_bf16x3.hip), errors those of the fp32 kernels -- "f16x2" = two fp16 planes per
    operand under one scale per tensor, three products on the fp16 MFMA (csrc/gemm_f16x2_256.hip; unit rows need no range
    management) -- or "f32" = the fp32 MFMA.
    ``fused`` (precision "f32x3", catalogues of more than two first blocks): everything after the first ``first_block``
    catalogue rows through the filter epilogue instead of score blocks (False: score blocks throughout, the round-5 form);
    ``q_chunk`` queries x ``c_chunk`` catalogue rows per filter launch (operands inside a descriptor's 2 GiB window, candidate
    buffer q_chunk x list capacity x 8 B), the lists merged -- and every query's threshold tightened -- after each."""
    if precision not in ("f32x3", "f32", "f16x2"):
        raise ValueError("precision must be 'f32x3', 'f16x2' or 'f32'")
    h2 = precision == "f16x2"
    x3 = precision == "f32x3" or h2                          # (the plane forms)
    cap = ops.knn_list_capacity()
    if not 1 <= k <= cap:
        raise ValueError("nearest_num must be in [1, %d]" % cap)
    nb, D = base.shape
    nq = queries.shape[0]
    if queries.shape[1] != D:
        raise ValueError("query / catalogue dimension mismatch")
    b_block = _round_up(min(b_block, _round_up(nb, 64)), 256 if x3 else 64)
    colm = 128 if h2 else 64 if x3 else 32
    B = _device_matrix(base, device, b_block, colm)
    same = queries is base
    Q = B if same else _device_matrix(queries, device, 1, colm)
    Dp = B.shape[1]
    if l2_norm:                                              # faiss_knn.py:99-104
        ops.l2norm_fwd(B[:nb], Dp, B)
        if not same:
            ops.l2norm_fwd(Q[:nq], Dp, Q)
    b_sq = torch.zeros(B.shape[0], dtype=torch.float32, device=device)
    ops.row_sqnorm(B[:nb], Dp, b_sq)
    if same:
        q_sq = b_sq
    else:
        q_sq = torch.zeros(nq, dtype=torch.float32, device=device)
        ops.row_sqnorm(Q[:nq], Dp, q_sq)
    best_d = torch.empty((nq, cap), dtype=torch.float32, device=device)
    best_i = torch.empty((nq, cap), dtype=torch.int32, device=device)
    q_block = min(q_block, nq)
    scores = torch.empty((q_block, b_block), dtype=torch.float32, device=device)
    if h2:
        B3, sb = _planes_h2(B, Dp)
        Q3, sq = (B3, sb) if same else _planes_h2(Q, Dp)
        osc = 1.0 / (sq * sb)
    elif x3:
        B3 = _planes(B, Dp)
        Q3 = B3 if same else _planes(Q, Dp)
    n_pad = B.shape[0]
    first_cols = n_pad
    use_filter = x3 and fused and first_block % b_block == 0 and first_block >= 4 * k and nb > 2 * first_block
    if use_filter:
        # a catalogue so long that a query would collect more than ~512 candidates behind a 32 768-row first block gets a
        # longer first block (k n / 512 rows: a 10 M-row catalogue sends 10 % of its rows through score blocks)
        first_cols = min(max(first_block, _round_up(k * nb // 512, b_block)), n_pad)
        use_filter = first_cols + 256 <= n_pad
        if not use_filter:
            first_cols = n_pad
    # queries and catalogue in chunks: a chunk's operands stay inside the 2 GiB window of a buffer descriptor, and the
    # candidate lists of a query chunk inside a bounded buffer
    q_chunk = _round_up(min(q_chunk, nq), q_block) if use_filter else nq
    c_chunk = _round_up(c_chunk, 256)
    lim = (2 ** 31) // (3 * Dp * 2) - 512
    q_chunk, c_chunk = min(q_chunk, lim // q_block * q_block), min(c_chunk, lim // 256 * 256)
    if use_filter:
        list_cap = 64
        expect = k * min(nb - first_cols, c_chunk) / float(first_cols)     # per catalogue chunk: merged and tau tightened after each
        while list_cap < 4 * expect:
            list_cap *= 2
        cnt = torch.zeros(q_chunk, dtype=torch.int32, device=device)
        cand = torch.empty((q_chunk, list_cap, 2), dtype=torch.int32, device=device)
        overflow = torch.zeros(1, dtype=torch.int32, device=device)
    for qs in range(0, nq, q_chunk):
        mq = min(q_chunk, nq - qs)
        for q0 in range(qs, qs + mq, q_block):
            m = min(q_block, qs + mq - q0)
            for c0 in range(0, first_cols, b_block):
                # scores[m, b_block] = Q[q0:q0+m] @ B[c0:c0+b_block]^T
                if h2:
                    ops.gemm_f16x2_nt(ops.BE_F32, Q3[q0:q0 + m], Dp, B3[c0:c0 + b_block], Dp, scores, m, b_block, Dp, osc)
                elif x3:
                    ops.gemm_bf16x3_nt(ops.BE_F32, Q3[q0:q0 + m], Dp, B3[c0:c0 + b_block], Dp, scores, m, b_block, Dp)
                else:
                    ops.fc_bwd_data(Q[q0:q0 + m], B[c0:c0 + b_block], None, scores, m, b_block, Dp)
                ops.knn_merge(scores, m, b_block, c0, nb, q_sq[q0:q0 + m], b_sq[c0:c0 + b_block], k,
                              best_d[q0:q0 + m], best_i[q0:q0 + m], first=(c0 == 0))
        if not use_filter:
            continue
        for c0 in range(first_cols, n_pad, c_chunk):
            nc = min(c_chunk, n_pad - c0)
            tau = best_d[qs:qs + mq, k - 1].contiguous()     # the k-th best so far: every merged chunk tightens it
            if h2:
                ops.knn_filter_h2(Q3[qs:qs + mq], Dp, B3[c0:c0 + nc], Dp, mq, nc, Dp, osc, q_sq[qs:qs + mq], b_sq[c0:c0 + nc], tau, c0,
                                  nb, cnt, cand, list_cap)
            else:
                ops.knn_filter_x3(Q3[qs:qs + mq], Dp, B3[c0:c0 + nc], Dp, mq, nc, Dp, q_sq[qs:qs + mq], b_sq[c0:c0 + nc], tau, c0, nb,
                                  cnt, cand, list_cap)
            ops.knn_merge_list(cand, cnt, list_cap, mq, k, best_d[qs:qs + mq], best_i[qs:qs + mq], overflow)
    if use_filter and int(overflow.item()):                  # (a sync; the export is not a step path)
        return knn_search(base, queries, k, l2_norm=l2_norm, device=device, q_block=q_block, b_block=b_block,
                          precision=precision, fused=False)
    I = best_i[:, :k].to(torch.int64)
    I[I == 0x7fffffff] = -1
    return best_d[:, :k].contiguous(), I


def calc_knn(embeddings, q_embeddings=None, nearest_num=51, l2_norm=True, M=80, efConstruction=64,
             efSearch=32, device="cuda:0", precision="f32x3"):
    """(D, I) ndarrays as faiss ``index.search(q, nearest_num)`` returns them
    (faiss_knn.py:128).  Like the reference, queries default to the catalogue."""
    emb = embeddings if torch.is_tensor(embeddings) else np.asarray(embeddings, dtype=np.float32)
    q = emb if q_embeddings is None else q_embeddings
    D, I = knn_search(emb, q, int(nearest_num), l2_norm=l2_norm, device=device, precision=precision)
    return D.cpu().numpy(), I.cpu().numpy()
