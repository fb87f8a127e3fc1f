"""Oracle: numpy restatement of the reference's embedding tower, triplet hinge
loss, their backward pass and the optimizers.  TEST INFRASTRUCTURE (see
oracle/__init__.py) -- the product never imports this.

Every function is written for an explicit ``dtype`` so the same code gives the
fp32 restatement (what TF1 computes) and its fp64 twin (the arbiter when two
fp32 summation orders disagree).

Parity status:
  * hinge loss  -- pinned by the hand-derived known answer of the reference's
    tests/test_losses.py:13-18 input (18.06 @ margin 0.1, 18.48 @ 0.8).
  * VNet tower, backward, Adam, LARS -- PARITY UNPINNED: the arithmetic lives in
    tensorflow-gpu==1.13.1 (README.md:21), which is not in /root/reference and
    not installable here; the reference holds no golden vector for them.  The
    restatement follows the reference call sites cited per function and TF 1.13
    op semantics, and is gradient-checked against torch CPU autograd in tests.

Reference citations are relative to /root/reference/.
"""
import numpy as np

L2_EPS = 1e-12          # tf.nn.l2_normalize default epsilon (models.py:58,61)
LRELU_ALPHA = 0.2       # tf.nn.leaky_relu default alpha (models.py:21)


# ----------------------------------------------------------------------------
# forward pieces
# ----------------------------------------------------------------------------
def l2_normalize(x, dtype=np.float32):
    """tf.nn.l2_normalize(x, axis=-1): x * rsqrt(max(sum(x^2), 1e-12)).

    models.py:58 (input) and models.py:61 (output).  Note: NOT x / max(|x|, eps).
    Returns (y, inv_norm[...,1])."""
    x = np.asarray(x, dtype=dtype)
    ss = np.sum(np.square(x), axis=-1, keepdims=True, dtype=dtype)
    inv = (dtype(1.0) / np.sqrt(np.maximum(ss, dtype(L2_EPS)))).astype(dtype)
    return (x * inv).astype(dtype), inv


def leaky_relu(x, alpha=LRELU_ALPHA):
    """tf.nn.leaky_relu: max(alpha*x, x) (models.py:21 default activation_fn)."""
    return np.maximum(x * x.dtype.type(alpha), x)


def fully_connected(x, W, b, alpha=LRELU_ALPHA):
    """slim.fully_connected with leaky_relu: lrelu(x @ W + b); W is [in, out]
    (models.py:19-30)."""
    return leaky_relu(x @ W + b, alpha)


def vnet_forward(x, W1, b1, W2, b2, dtype=np.float32, alpha=LRELU_ALPHA):
    """VNet.create_model (models.py:46-62).

    x [R,F] -> l2norm -> FC(H)+lrelu -> FC(D)+lrelu -> l2norm.
    Returns the reference's dict keys plus the intermediates backward needs."""
    x = np.asarray(x, dtype)
    W1, b1, W2, b2 = (np.asarray(a, dtype) for a in (W1, b1, W2, b2))
    x_hat, _ = l2_normalize(x, dtype)
    layer_1 = fully_connected(x_hat, W1, b1, alpha).astype(dtype)
    layer_2 = fully_connected(layer_1, W2, b2, alpha).astype(dtype)
    l2, inv2 = l2_normalize(layer_2, dtype)
    return {"x_hat": x_hat, "layer_1": layer_1, "layer_2": layer_2,
            "l2_norm": l2, "inv_norm_2": inv2}


def hinge_loss(triplets, margin=0.1, dtype=np.float32):
    """HingeLoss.calculate_loss (losses.py:20-49).  triplets [B,3,D].

    Squared L2 distances; outputs keep the split axis ([B,1]) like tf.split."""
    t = np.asarray(triplets, dtype)
    anchors, positives, negatives = t[:, 0:1, :], t[:, 1:2, :], t[:, 2:3, :]
    pos_dist = np.sum(np.square(anchors - positives), axis=-1, dtype=dtype)
    neg_dist = np.sum(np.square(anchors - negatives), axis=-1, dtype=dtype)
    hinge_dist = np.maximum(pos_dist - neg_dist + dtype(margin), dtype(0))
    hinge_loss_ = np.mean(hinge_dist, dtype=dtype)
    return {"hinge_loss": dtype(hinge_loss_), "anchors": anchors,
            "positives": positives, "negatives": negatives,
            "pos_dist": pos_dist, "neg_dist": neg_dist,
            "hinge_dist": hinge_dist}


def calc_var(triplets, dtype=np.float32):
    """train.py:67-71: mean over everything of (t - mean over [batch,channel])^2."""
    t = np.asarray(triplets, dtype)
    mean = np.mean(t, axis=(0, 1), dtype=dtype)
    return dtype(np.mean(np.square(t - mean), dtype=dtype))


# ----------------------------------------------------------------------------
# indexed triplets (build-defined extension: in-batch negatives; PARITY UNPINNED,
# no reference counterpart -- SURVEY.md section 8a "no reference row")
# ----------------------------------------------------------------------------
def hinge_loss_indexed(E, tri, valid, margin, dtype=np.float32):
    """Loss over triplets given as row indices into E [R,D].

    tri int[B,3] = (anchor row, positive row, negative row); valid bool[B]
    masks triplets out (hinge := 0, no gradient) but they still count in the
    mean's denominator B."""
    E = np.asarray(E, dtype)
    a, p, n = E[tri[:, 0]], E[tri[:, 1]], E[tri[:, 2]]
    pos = np.sum(np.square(a - p), axis=-1, dtype=dtype)
    neg = np.sum(np.square(a - n), axis=-1, dtype=dtype)
    hinge = np.maximum(pos - neg + dtype(margin), dtype(0))
    hinge = np.where(valid, hinge, dtype(0)).astype(dtype)
    loss = dtype(np.sum(hinge, dtype=dtype) / dtype(len(tri)))
    return {"hinge_loss": loss, "pos_dist": pos, "neg_dist": neg,
            "hinge_dist": hinge}


def hinge_loss_indexed_backward(E, tri, valid, margin, dtype=np.float32):
    """dLoss/dE for hinge_loss_indexed, accumulated over shared rows."""
    E = np.asarray(E, dtype)
    B = len(tri)
    a, p, n = E[tri[:, 0]], E[tri[:, 1]], E[tri[:, 2]]
    pos = np.sum(np.square(a - p), axis=-1, dtype=dtype)
    neg = np.sum(np.square(a - n), axis=-1, dtype=dtype)
    act = ((pos - neg + dtype(margin)) >= 0) & valid
    s = (act.astype(dtype) * dtype(2.0) / dtype(B))[:, None]
    dE = np.zeros_like(E)
    np.add.at(dE, tri[:, 0], s * (n - p))
    np.add.at(dE, tri[:, 1], -s * (a - p))
    np.add.at(dE, tri[:, 2], s * (a - n))
    return dE.astype(dtype)


# ----------------------------------------------------------------------------
# backward (what optimizer.compute_gradients builds, train.py:141)
# ----------------------------------------------------------------------------
def hinge_loss_backward(triplets, margin, dtype=np.float32):
    """d mean(hinge) / d triplets  [B,3,D].

    TF's MaximumGrad routes the gradient to x where x >= y, so a triplet with
    pos-neg+margin == 0 exactly is still 'active'."""
    t = np.asarray(triplets, dtype)
    B = t.shape[0]
    a, p, n = t[:, 0, :], t[:, 1, :], t[:, 2, :]
    pos = np.sum(np.square(a - p), axis=-1, dtype=dtype)
    neg = np.sum(np.square(a - n), axis=-1, dtype=dtype)
    act = ((pos - neg + dtype(margin)) >= 0).astype(dtype)[:, None]
    s = act * dtype(2.0) / dtype(B)
    d = np.empty_like(t)
    d[:, 0, :] = s * (n - p)          # 2(a-p) - 2(a-n)
    d[:, 1, :] = -s * (a - p)
    d[:, 2, :] = s * (a - n)
    return d


def l2_normalize_backward(z, inv, g, dtype=np.float32):
    """Gradient of y = z * rsqrt(max(sum z^2, eps)) wrt z.

    Where the sum is above eps: dz = inv * (g - y * sum(y*g)); where it is
    clamped the factor is a constant: dz = inv * g."""
    z = np.asarray(z, dtype)
    g = np.asarray(g, dtype)
    y = z * inv
    ss = np.sum(np.square(z), axis=-1, keepdims=True, dtype=dtype)
    dot = np.sum(y * g, axis=-1, keepdims=True, dtype=dtype)
    full = inv * (g - y * dot)
    clamped = inv * g
    return np.where(ss > dtype(L2_EPS), full, clamped).astype(dtype)


def leaky_relu_backward(y_post, g, alpha=LRELU_ALPHA):
    """LeakyReluGrad: g where features > 0 else alpha*g.  The sign of the
    post-activation equals the sign of the pre-activation (alpha > 0)."""
    return np.where(y_post > 0, g, g * g.dtype.type(alpha))


def vnet_backward(fwd, W2, dE, dtype=np.float32, alpha=LRELU_ALPHA):
    """Backprop dE [R,D] (grad wrt the l2-normalised output) to the four
    parameter tensors.  The input is a placeholder (train.py:265): no dX."""
    W2 = np.asarray(W2, dtype)
    dZ2 = l2_normalize_backward(fwd["layer_2"], fwd["inv_norm_2"], dE, dtype)
    dZ2 = leaky_relu_backward(fwd["layer_2"], dZ2, alpha).astype(dtype)
    dW2 = (fwd["layer_1"].T @ dZ2).astype(dtype)
    db2 = np.sum(dZ2, axis=0, dtype=dtype)
    dH1 = (dZ2 @ W2.T).astype(dtype)
    dZ1 = leaky_relu_backward(fwd["layer_1"], dH1, alpha).astype(dtype)
    dW1 = (fwd["x_hat"].T @ dZ1).astype(dtype)
    db1 = np.sum(dZ1, axis=0, dtype=dtype)
    return {"dW1": dW1, "db1": db1, "dW2": dW2, "db2": db2,
            "dZ2": dZ2, "dZ1": dZ1}


def train_step_grads(x, params, margin, dtype=np.float32):
    """One reference step's loss + gradients for x [3B,F] in a,p,n row order
    (train.py:313 reshape, :128 reshape back, :130 loss, :141 gradients;
    regularization_penalty=0 and clip_gradient_norm=0 as in train.py:221-222)."""
    W1, b1, W2, b2 = params
    fwd = vnet_forward(x, W1, b1, W2, b2, dtype)
    D = fwd["l2_norm"].shape[-1]
    trip = fwd["l2_norm"].reshape(-1, 3, D)
    loss = hinge_loss(trip, margin, dtype)
    dE = hinge_loss_backward(trip, margin, dtype).reshape(-1, D)
    grads = vnet_backward(fwd, W2, dE, dtype)
    return fwd, loss, grads


# ----------------------------------------------------------------------------
# optimizers + LR schedule (train.py:108-125,146)
# ----------------------------------------------------------------------------
def exponential_decay(base_lr, global_step, decay_steps, decay_rate,
                      staircase=True):
    """tf.train.exponential_decay (train.py:108-113)."""
    p = global_step / float(decay_steps)
    if staircase:
        p = np.floor(p)
    return base_lr * (decay_rate ** p)


def adam_step(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              dtype=np.float32):
    """tf.train.AdamOptimizer update at (1-based) step t -- the arithmetic of
    TF 1.13's ApplyAdam functor, epsilon OUTSIDE the bias correction:
        alpha = lr * sqrt(1-b2^t) / (1-b1^t)
        m += (g - m)*(1-b1) ; v += (g*g - v)*(1-b2) ; w -= alpha*m/(sqrt(v)+eps)
    with (1-b1), (1-b2) formed in ``dtype`` (in fp32 1-0.999 = 0.0010000467).
    Returns new (w, m, v).  PARITY UNPINNED (TF 1.13 source recalled)."""
    w, g, m, v = (np.asarray(a, dtype) for a in (w, g, m, v))
    alpha = dtype(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    omb1 = dtype(1) - dtype(beta1)
    omb2 = dtype(1) - dtype(beta2)
    m = (m + (g - m) * omb1).astype(dtype)
    v = (v + (g * g - v) * omb2).astype(dtype)
    w = (w - (m * alpha) / (np.sqrt(v) + dtype(eps))).astype(dtype)
    return w, m, v


def lars_step(w, g, acc, lr, momentum=0.9, weight_decay=1e-4, eeta=1e-3,
              epsilon=0.0, dtype=np.float32):
    """tf.contrib.opt.LARSOptimizer (train.py:354) per-variable update with the
    contrib defaults -- recalled from TF 1.13 source, PARITY UNPINNED:
        trust = eeta*|w| / (|g| + wd*|w| + eps)   (1.0 if |w|==0 or |g|==0)
        scaled_lr = lr*trust ; g <- g + wd*w
        acc <- momentum*acc + scaled_lr*g ; w <- w - acc
    (the scaled lr is folded into the momentum accumulator, as MomentumOptimizer
    does with its learning_rate)."""
    w, g, acc = (np.asarray(a, dtype) for a in (w, g, acc))
    w_norm = np.sqrt(np.sum(np.square(w, dtype=np.float64)))
    g_norm = np.sqrt(np.sum(np.square(g, dtype=np.float64)))
    if w_norm > 0 and g_norm > 0:
        trust = eeta * w_norm / (g_norm + weight_decay * w_norm + epsilon)
    else:
        trust = 1.0
    scaled_lr = dtype(lr * trust)
    g = (g + dtype(weight_decay) * w).astype(dtype)
    acc = (dtype(momentum) * acc + scaled_lr * g).astype(dtype)
    w = (w - acc).astype(dtype)
    return w, acc


def clip_by_norm(g, clip_norm, dtype=np.float32):
    """tf.clip_by_norm (train.py:47-64 applies it per variable): g * clip / max(|g|_2, clip)."""
    g = np.asarray(g, dtype)
    n = np.sqrt(np.sum(np.square(g), dtype=dtype))
    return (g * (dtype(clip_norm) / np.maximum(n, dtype(clip_norm)))).astype(dtype)


def regularized_grads(grads, params, regularization_penalty, l2_penalty=1e-8, dtype=np.float32):
    """final_loss = regularization_penalty * reg_loss + loss (train.py:133-139) with reg_loss =
    sum over the weight matrices of slim.l2_regularizer(l2_penalty)(W) = l2_penalty*|W|^2/2
    (models.py:28; biases carry no regulariser): dW += penalty*l2_penalty*W.  Returns
    (grads incl. the term as a dict like vnet_backward's, reg_loss)."""
    W1, _, W2, _ = [np.asarray(p, dtype) for p in params]
    s = dtype(regularization_penalty) * dtype(l2_penalty)
    out = dict(grads)
    out["dW1"] = (np.asarray(grads["dW1"], dtype) + s * W1).astype(dtype)
    out["dW2"] = (np.asarray(grads["dW2"], dtype) + s * W2).astype(dtype)
    reg = dtype(l2_penalty) * (np.sum(np.square(W1), dtype=dtype) + np.sum(np.square(W2), dtype=dtype)) / dtype(2)
    return out, dtype(reg)


def momentum_step(w, g, acc, lr, momentum=0.9, use_nesterov=True, dtype=np.float32):
    """tf.train.MomentumOptimizer(lr, momentum=0.9, use_nesterov=True) (train.py:115-116), TF's
    ApplyMomentum functor: acc = acc*momentum + g; w -= g*lr + acc*momentum*lr (Nesterov) or
    acc*lr.  Restated from the op's definition; parity unpinned (no TF here)."""
    w, g, acc = (np.asarray(a, dtype) for a in (w, g, acc))
    acc = (acc * dtype(momentum) + g).astype(dtype)
    if use_nesterov:
        w = w - (g * dtype(lr) + acc * dtype(momentum) * dtype(lr))
    else:
        w = w - acc * dtype(lr)
    return w.astype(dtype), acc


def xavier_uniform(rng, fan_in, fan_out, dtype=np.float32):
    """slim's default weights_initializer (xavier, uniform): U(+-sqrt(6/(in+out)))."""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


# ----------------------------------------------------------------------------
# semi-hard negative mining (BASELINE config 2) -- build-defined, PARITY UNPINNED:
# the reference only draws uniform random negatives.  Rule (the one of
# tf.contrib.losses.metric_learning.triplet_semihard_loss, on squared distances):
# for anchor i with positive distance d_p, over every OTHER embedded row c of the
# batch whose video is neither the anchor's nor the positive's,
#   pick the closest c with d(a,c) > d_p ("semi-hard / outside"),
#   else (no such c) the farthest eligible c;  ties -> smallest row index;
#   no eligible c at all -> -1 (triplet masked).
# ----------------------------------------------------------------------------
def semihard_select(E, rows, dtype=np.float64):
    """E [2B,D] embeddings (row 2i anchor, 2i+1 positive), rows int[2B] video ids.
    Returns (neg_row int32[B], dist float[B,2B]) with dist = squared distances of
    each anchor to every row (the selection's input, for tolerance-aware checks)."""
    E = np.asarray(E, dtype)
    rows = np.asarray(rows)
    B = len(rows) // 2
    A = E[0::2]
    sq = np.sum(E * E, axis=1)
    dist = sq[0::2, None] + sq[None, :] - 2.0 * (A @ E.T)
    d_p = dist[np.arange(B), 2 * np.arange(B) + 1]
    eligible = (rows[None, :] != rows[0::2, None]) & (rows[None, :] != rows[1::2, None])
    outside = eligible & (dist > d_p[:, None])
    neg = np.full(B, -1, dtype=np.int32)
    for i in range(B):
        if outside[i].any():
            d = np.where(outside[i], dist[i], np.inf)
            neg[i] = int(np.argmin(d))
        elif eligible[i].any():
            d = np.where(eligible[i], dist[i], -np.inf)
            neg[i] = int(np.argmax(d))
    return neg, dist


def semihard_triplets(neg_row):
    """tri int32[B,3] row indices + valid mask for hinge_loss_indexed."""
    B = len(neg_row)
    valid = neg_row >= 0
    tri = np.stack([2 * np.arange(B), 2 * np.arange(B) + 1, np.where(valid, neg_row, 0)], axis=1)
    return tri.astype(np.int32), valid


# ----------------------------------------------------------------------------
# fusion towers (next-row N4): MultiplyNet / MlpNet / ResNet, models.py:65-157.
# Input = visual (first 1500 columns) ++ doc features; every fully_connected has
# leaky_relu and bias_init 0.1.  PARITY UNPINNED like VNet (TF 1.13 not available).
# ----------------------------------------------------------------------------
FUSION_LAYERS = {
    "MultiplyNet": ("layer_visual_1", "layer_visual_2", "layer_doc_1", "layer_doc_2"),
    "MlpNet": ("layer_visual_1", "layer_visual_2", "layer_doc_1", "layer_doc_2",
               "layer_fusion_1", "layer_fusion_2"),
    "ResNet": ("layer_visual_1", "layer_visual_2", "layer_doc_1", "layer_doc_2",
               "layer_fusion_1", "layer_fusion_2"),
    # models.py:205-243: a deep and a shallow branch per modality, four cross products
    "ResNetV2": ("layer_visual_1_1", "layer_visual_1_2", "layer_visual_2_1", "layer_doc_1_1",
                 "layer_doc_1_2", "layer_doc_2_1", "layer_fusion_1", "layer_fusion_2"),
}


def fusion_layer_shapes(net, visual=1500, doc=128, hidden_v=5000, hidden_d=400, out=256, mlp_hidden=600):
    """(fan_in, fan_out) per layer name (models.py:79-88,106-120,137-153)."""
    s = {"layer_visual_1": (visual, hidden_v), "layer_visual_2": (hidden_v, out),
         "layer_doc_1": (doc, hidden_d), "layer_doc_2": (hidden_d, out)}
    if net == "MlpNet":
        s["layer_fusion_1"], s["layer_fusion_2"] = (out, mlp_hidden), (mlp_hidden, out)
    elif net == "ResNet":
        s["layer_fusion_1"], s["layer_fusion_2"] = (out, out), (out, out)
    elif net == "ResNetV2":
        s = {"layer_visual_1_1": (visual, hidden_v), "layer_visual_1_2": (hidden_v, out),
             "layer_visual_2_1": (visual, out), "layer_doc_1_1": (doc, hidden_d),
             "layer_doc_1_2": (hidden_d, out), "layer_doc_2_1": (doc, out),
             "layer_fusion_1": (out, out), "layer_fusion_2": (out, out)}
    return {k: s[k] for k in FUSION_LAYERS[net]}


def fusion_forward(net, x, P, visual=1500, dtype=np.float64):
    """P: name -> (W, b).  Returns dict of every intermediate + 'l2_norm'."""
    x = np.asarray(x, dtype)
    fc = lambda a, n: fully_connected(a, np.asarray(P[n][0], dtype), np.asarray(P[n][1], dtype)).astype(dtype)
    t = {}
    t["xv"], _ = l2_normalize(x[:, :visual], dtype)
    t["xd"], _ = l2_normalize(x[:, visual:], dtype)
    if net == "ResNetV2":
        # models.py:219-238.  f1+f2+f3+f4 = (v12+v21)*(d12+d21), so layer_res_1 is ResNet's
        # a*b + a + b with a, b the SUMS of the deep and the shallow branch
        t["v11"] = fc(t["xv"], "layer_visual_1_1"); t["v12"] = fc(t["v11"], "layer_visual_1_2")
        t["v21"] = fc(t["xv"], "layer_visual_2_1")
        t["d11"] = fc(t["xd"], "layer_doc_1_1"); t["d12"] = fc(t["d11"], "layer_doc_1_2")
        t["d21"] = fc(t["xd"], "layer_doc_2_1")
        t["r1"] = (t["v12"] * t["d12"] + t["v12"] * t["d21"] + t["v21"] * t["d12"] + t["v21"] * t["d21"]
                   + t["v12"] + t["v21"] + t["d12"] + t["d21"])
        t["f1"] = fc(t["r1"], "layer_fusion_1"); t["r2"] = t["r1"] + t["f1"]
        t["f2"] = fc(t["r2"], "layer_fusion_2"); t["pre_norm"] = t["r2"] + t["f2"]
        t["l2_norm"], t["inv"] = l2_normalize(t["pre_norm"], dtype)
        return t
    t["v1"] = fc(t["xv"], "layer_visual_1"); t["v2"] = fc(t["v1"], "layer_visual_2")
    t["d1"] = fc(t["xd"], "layer_doc_1"); t["d2"] = fc(t["d1"], "layer_doc_2")
    if net == "MultiplyNet":
        t["pre_norm"] = t["v2"] * t["d2"]
    elif net == "MlpNet":
        t["fu"] = t["v2"] * t["d2"]
        t["f1"] = fc(t["fu"], "layer_fusion_1"); t["f2"] = fc(t["f1"], "layer_fusion_2")
        t["pre_norm"] = t["f2"]
    else:
        t["r1"] = t["v2"] * t["d2"] + t["v2"] + t["d2"]
        t["f1"] = fc(t["r1"], "layer_fusion_1"); t["r2"] = t["r1"] + t["f1"]
        t["f2"] = fc(t["r2"], "layer_fusion_2"); t["pre_norm"] = t["r2"] + t["f2"]
    t["l2_norm"], t["inv"] = l2_normalize(t["pre_norm"], dtype)
    return t


def fusion_backward(net, t, P, dE, dtype=np.float64):
    """Gradients name -> (dW, db) for dE = d loss / d l2_norm."""
    W = {k: np.asarray(v[0], dtype) for k, v in P.items()}
    g = {}

    def fc_bwd(name, x_in, y_post, d_post):
        d_pre = leaky_relu_backward(y_post, d_post)
        g[name] = (x_in.T @ d_pre, d_pre.sum(0))
        return d_pre @ W[name].T                     # wrt the layer's input (post-activation of its producer)

    d = l2_normalize_backward(t["pre_norm"], t["inv"], np.asarray(dE, dtype), dtype)
    if net == "ResNetV2":
        d_r2 = d + fc_bwd("layer_fusion_2", t["r2"], t["f2"], d)
        d_r1 = d_r2 + fc_bwd("layer_fusion_1", t["r1"], t["f1"], d_r2)
        d_v = d_r1 * (t["d12"] + t["d21"] + 1.0)              # wrt v12 and wrt v21 alike
        d_d = d_r1 * (t["v12"] + t["v21"] + 1.0)
        fc_bwd("layer_visual_1_1", t["xv"], t["v11"], fc_bwd("layer_visual_1_2", t["v11"], t["v12"], d_v))
        fc_bwd("layer_visual_2_1", t["xv"], t["v21"], d_v)
        fc_bwd("layer_doc_1_1", t["xd"], t["d11"], fc_bwd("layer_doc_1_2", t["d11"], t["d12"], d_d))
        fc_bwd("layer_doc_2_1", t["xd"], t["d21"], d_d)
        return g
    if net == "MultiplyNet":
        dfu, res = d, 0.0
    elif net == "MlpNet":
        d_f1 = fc_bwd("layer_fusion_2", t["f1"], t["f2"], d)
        dfu, res = fc_bwd("layer_fusion_1", t["fu"], t["f1"], d_f1), 0.0
    else:
        d_r2 = d + fc_bwd("layer_fusion_2", t["r2"], t["f2"], d)
        dfu, res = d_r2 + fc_bwd("layer_fusion_1", t["r1"], t["f1"], d_r2), 1.0
    d_v2, d_d2 = dfu * (t["d2"] + res), dfu * (t["v2"] + res)
    d_v1 = fc_bwd("layer_visual_2", t["v1"], t["v2"], d_v2)
    fc_bwd("layer_visual_1", t["xv"], t["v1"], d_v1)
    d_d1 = fc_bwd("layer_doc_2", t["d1"], t["d2"], d_d2)
    fc_bwd("layer_doc_1", t["xd"], t["d1"], d_d1)
    return g
