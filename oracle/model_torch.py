"""CPU restatement (PyTorch-CPU, float32 or float64) of the reference's Keras graph, losses and
optimizer step.  TEST INFRASTRUCTURE (see oracle/__init__.py).

"PARITY UNPINNED": tensorflow / keras / keras_resnet are not installed here and the reference holds
no tests or golden tensors for this path, so this file follows the reference source line by line
(cited below, relative to /root/reference) plus the published behaviour of the pinned third-party
packages -- keras 2.3.1 (Conv2D 'same', binary_crossentropy, Adam/clipnorm), keras-resnet 0.1.0
(ResNet50 topology, Caffe-style stride placement, frozen BN eps=1e-5), tensorflow 2.1.1
(tf.image.resize NEAREST with half-pixel centres).  It is checked by analytic known-answer tests
(tests/test_oracle_model.py).

Graph:  models/resnet.py:79-110  -> keras_resnet ResNet50(include_top=False, freeze_bn=True), C3..C5
        models/retinanet.py:180-214  __create_sparceFPN
        models/retinanet.py:101-131 / 9-54 / 57-98  regression / classification / mask sub-models
        models/retinanet.py:224-229, 296-299  concat over levels, outputs ['3Dbox', 'cls', 'mask']
Weights are a dict keyed by layer name: '<conv>/kernel' HWIO, '<conv>/bias', '<bn>/{gamma,beta,mean,var}'.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # keras_resnet.layers.BatchNormalization(epsilon=1e-5)
RESNET50_BLOCKS = [3, 4, 6, 3]


def _t(a, dtype):
    if torch.is_tensor(a):
        return a.to(dtype)
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def tf_same_pad(n_in, k, s):
    """TensorFlow 'SAME': total = max((ceil(in/s)-1)*s + k - in, 0), before = total // 2."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2


def conv2d(x, w_hwio, bias=None, stride=1, padding="same"):
    """x: NCHW.  padding: 'same' (TF rule), or an int (ZeroPadding2D(p) followed by 'valid')."""
    w = w_hwio.permute(3, 2, 0, 1)  # HWIO -> OIHW
    kh, kw = w.shape[2], w.shape[3]
    if padding == "same":
        pt, pb = tf_same_pad(x.shape[2], kh, stride)
        pl, pr = tf_same_pad(x.shape[3], kw, stride)
    else:
        pt = pb = pl = pr = int(padding)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, bias, stride=stride)


def frozen_bn(x, W, name, dtype):
    """keras_resnet BatchNormalization(freeze=True): inference-mode affine (models/resnet.py:87)."""
    g, b = _t(W[name + "/gamma"], dtype), _t(W[name + "/beta"], dtype)
    m, v = _t(W[name + "/mean"], dtype), _t(W[name + "/var"], dtype)
    scale = g / torch.sqrt(v + BN_EPS)
    shift = b - m * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def upsample_like(src, target):
    """layers/_misc.py:96-109 -> backend/tf_backend.py:28-35: tf.image.resize(NEAREST), TF 2.1:
    src index = min(floor((dst + 0.5) * in/out), in - 1), scale in float32."""
    def idx(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        i = np.floor((np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale).astype(np.int64)
        return torch.as_tensor(np.minimum(i, n_in - 1))
    iy = idx(src.shape[2], target.shape[2])
    ix = idx(src.shape[3], target.shape[3])
    return src[:, :, iy][:, :, :, ix]


def _relu(z, masks, name, level=None):
    """ReLU, or -- when `masks` (name -> 0/1 tensor like z, or a list of them per pyramid level) is given -- multiplication by
    a FIXED 0/1 pattern taken from another evaluation of the same graph.  The loss is piecewise smooth; two evaluations in
    different arithmetic may put a pre-activation that is zero to rounding on different sides of the kink, after which
    their gradients differ by O(1e-2) for a reason that has nothing to do with either implementation.  With the pattern
    pinned, both differentiate the SAME smooth piece and can be compared tightly (tests only)."""
    if masks is None:
        return F.relu(z)
    m = masks[name]
    if level is not None:
        m = m[level]
    assert m.shape == z.shape, (name, level, tuple(m.shape), tuple(z.shape))
    return z * m.to(z.dtype)


def resnet50(x, W, dtype, blocks=None, relu_masks=None):
    """keras_resnet.models.ResNet50(include_top=False, freeze_bn=True) -> [C2, C3, C4, C5].
    blocks = [3, 4, 23, 3] gives the ResNet-101 variant (keras_resnet numerical_names [F, T, T, F]: blocks of
    stages 3 and 4 are named 'a', 'b1', 'b2', ...)."""
    blocks = blocks or RESNET50_BLOCKS
    numerical = [False, True, True, False] if list(blocks) != RESNET50_BLOCKS else [False] * 4
    P = lambda n: _t(W[n + "/kernel"], dtype)
    y = conv2d(x, P("conv1"), None, 2, 3)                      # ZeroPadding2D(3) + 7x7/2 valid, no bias
    y = _relu(frozen_bn(y, W, "bn_conv1", dtype), relu_masks, "conv1")
    pt, pb = tf_same_pad(y.shape[2], 3, 2)                     # MaxPooling2D(3, 2, 'same')
    pl, pr = tf_same_pad(y.shape[3], 3, 2)
    y = F.max_pool2d(F.pad(y, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
    outs = []
    for stage, n_blocks in enumerate(blocks):
        for block in range(n_blocks):
            sc = str(stage + 2)
            bc = ("b%d" % block) if (block > 0 and numerical[stage]) else chr(ord("a") + block)
            stride = 1 if (block != 0 or stage == 0) else 2  # Caffe style: stride on the first 1x1
            nm = lambda br: ("res%s%s_branch%s" % (sc, bc, br), "bn%s%s_branch%s" % (sc, bc, br))
            c, b = nm("2a")
            z = _relu(frozen_bn(conv2d(y, P(c), None, stride, 0), W, b, dtype), relu_masks, c)
            c, b = nm("2b")
            z = _relu(frozen_bn(conv2d(z, P(c), None, 1, 1), W, b, dtype), relu_masks, c)   # ZeroPadding2D(1) + valid
            c, b = nm("2c")
            z = frozen_bn(conv2d(z, P(c), None, 1, 0), W, b, dtype)
            if block == 0:
                c, b = nm("1")
                short = frozen_bn(conv2d(y, P(c), None, stride, 0), W, b, dtype)
            else:
                short = y
            y = _relu(z + short, relu_masks, "res%s%s" % (sc, bc))
        outs.append(y)
    return outs


def sparse_fpn(C3, C4, C5, W, dtype):
    """models/retinanet.py:180-214 (__create_sparceFPN).  All convs biased, no activation."""
    def cv(name, x, k, s=1):
        return conv2d(x, _t(W[name + "/kernel"], dtype), _t(W[name + "/bias"], dtype), s, "same")
    L3 = cv("fpn_lat3", C3, 1); L4 = cv("fpn_lat4", C4, 1); L5 = cv("fpn_lat5", C5, 1)      # :192-194
    U5 = upsample_like(L5, L4); U4 = upsample_like(L4, L3)                                  # :196-197
    M4 = cv("fpn_mid4", U5 + L4, 3)                                                         # :198-199
    M3 = cv("fpn_mid3", U4 + L3, 3)                                                         # :200-201
    D3 = cv("fpn_down3", M3, 3, 2)                                                          # :202
    P3 = cv("P3", M3 + L3, 3)                                                               # :203-204
    F4 = D3 + M4                                                                            # :206
    D4 = cv("fpn_down4", M4, 3, 2)                                                          # :207
    P4 = cv("P4", F4 + L4, 3)                                                               # :208-209
    P5 = cv("P5", D4 + L5, 3)                                                               # :211-212
    return P3, P4, P5


def pyramid_features(C3, C4, C5, W, dtype, with_p6p7=True, relu_masks=None):
    """models/retinanet.py:134-157 (__create_pyramid_features, P3..P7) / :160-177 (__create_FPN, P3..P5)."""
    def cv(name, x, k, s=1):
        return conv2d(x, _t(W[name + "/kernel"], dtype), _t(W[name + "/bias"], dtype), s, "same")
    P5 = cv("C5_reduced", C5, 1)                                        # :135
    U5 = upsample_like(P5, C4)                                          # :136
    P5 = cv("P5_con", P5, 3)                                            # :137
    P4 = cv("C4_reduced", C4, 1) + U5                                   # :140-141
    U4 = upsample_like(P4, C3)                                          # :142
    P4 = cv("P4_con", P4, 3)                                            # :143
    P3 = cv("P3_con", cv("C3_reduced", C3, 1) + U4, 3)                  # :146-148
    if not with_p6p7:
        return [P3, P4, P5]
    P6 = cv("P6_con", C5, 3, 2)                                         # :151 "3x3 stride-2 conv on C5"
    P7 = cv("P7_con", _relu(P6, relu_masks, "P6_relu"), 3, 2)                                 # :154-155
    return [P3, P4, P5, P6, P7]


def head(prefix, feat, W, dtype, n_values, relu_masks=None, level=0):
    """4 x [3x3 conv + ReLU] + 3x3 conv, then Reshape((-1, n_values)) on NHWC."""
    y = feat
    for i in range(4):
        n = "%s_conv%d" % (prefix, i)
        y = _relu(conv2d(y, _t(W[n + "/kernel"], dtype), _t(W[n + "/bias"], dtype)), relu_masks, n, level)
    y = conv2d(y, _t(W[prefix + "_out/kernel"], dtype), _t(W[prefix + "_out/bias"], dtype))
    return y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, n_values)


def forward(W, x_nhwc, num_classes, dtype=torch.float32, blocks=None, return_features=False, pyramid="sparse", relu_masks=None):
    """x_nhwc: (B,H,W,3) preprocessed image batch.  Returns dict with '3Dbox', 'cls', 'mask' (Keras
    outputs: cls/mask are probabilities) plus the pre-sigmoid logits."""
    x = _t(x_nhwc, dtype).permute(0, 3, 1, 2)
    C2, C3, C4, C5 = resnet50(x, W, dtype, blocks, relu_masks)
    if pyramid == "sparse":
        feats = list(sparse_fpn(C3, C4, C5, W, dtype))
    else:
        feats = pyramid_features(C3, C4, C5, W, dtype, with_p6p7=(pyramid == "p3p7"), relu_masks=relu_masks)
    P3, P4, P5 = feats[:3]
    rm = relu_masks
    reg = torch.cat([head("reg", f, W, dtype, 16, rm, l) for l, f in enumerate(feats)], dim=1)
    cls_logit = torch.cat([head("cls", f, W, dtype, num_classes, rm, l) for l, f in enumerate(feats)], dim=1)
    mask_logit = head("mask", P3, W, dtype, num_classes, rm, 0)
    out = {"3Dbox": reg, "cls": torch.sigmoid(cls_logit), "mask": torch.sigmoid(mask_logit),
           "cls_logit": cls_logit, "mask_logit": mask_logit}
    if return_features:
        out.update({"C2": C2, "C3": C3, "C4": C4, "C5": C5, "P3": P3, "P4": P4, "P5": P5})
    return out


# ------------------------------------------------------------------------------------------ losses
def keras_binary_crossentropy(target, output):
    """keras 2.3.1 tensorflow_backend.binary_crossentropy(from_logits=False): clip to
    [eps, 1-eps] (eps = 1e-7), convert to logits, tf.nn.sigmoid_cross_entropy_with_logits."""
    eps = 1e-7
    o = torch.clamp(output, eps, 1 - eps)
    logit = torch.log(o / (1 - o))
    return torch.clamp(logit, min=0) - logit * target + torch.log1p(torch.exp(-torch.abs(logit)))


def focal(y_true, y_pred, alpha=0.25, gamma=2.0):
    """losses.py:22-68.  y_true (B,N,C+1) with state in the last column; y_pred probabilities."""
    labels = y_true[:, :, :-1]
    state = y_true[:, :, -1]
    keep = state != -1
    labels = labels[keep]
    p = y_pred[keep]
    alpha_f = torch.where(labels == 1, torch.full_like(labels, alpha), torch.full_like(labels, 1 - alpha))
    fw = torch.where(labels == 1, 1 - p, p)
    fw = alpha_f * fw ** gamma
    cls_loss = fw * keras_binary_crossentropy(labels, p)
    normalizer = max(1.0, float((state == 1).sum()))
    return cls_loss.sum() / normalizer


_ORTH = [(0, 6, 2, 4), (0, 6, 8, 14), (0, 2, 6, 4), (0, 2, 8, 10), (0, 8, 2, 10), (0, 8, 6, 14),
         (12, 10, 14, 8), (12, 10, 4, 2), (12, 4, 10, 2), (12, 4, 14, 6), (12, 14, 4, 6), (12, 14, 10, 8)]


def _orth_features(r):
    """losses.py:338-362: x1,y1,...,x12,y12 with (a,b,c,d) -> (r[a]-r[b]) - (r[c]-r[d])."""
    f = []
    for (a, b, c, d) in _ORTH:
        f.append((r[:, a] - r[:, b]) - (r[:, c] - r[:, d]))
        f.append((r[:, a + 1] - r[:, b + 1]) - (r[:, c + 1] - r[:, d + 1]))
    return torch.stack(f, dim=1)


def orthogonal_l1(y_true, y_pred, weight=0.125, sigma=3.0, kink_ref=None):
    """losses.py:321-408.
    kink_ref (tests; like relu_masks): predictions of ANOTHER evaluation of the same network (the engine's box output).  The
    24 edge-difference terms enter through abs(), whose derivative jumps at 0; with a few hundred positives ONE term that is zero
    to rounding, and lands on different sides in two arithmetics, moves the gradient of the whole regression head by ~7e-4.
    With kink_ref the sign of every such term is taken from that evaluation, i.e. both sides differentiate the same smooth
    piece (the smooth-L1 part needs nothing: its derivative is continuous at 0 and at the knee)."""
    sigma_sq = sigma ** 2
    target = y_true[:, :, :-1]
    state = y_true[:, :, -1]
    pos = state == 1
    r = y_pred[pos]
    t = target[pos]
    diff = torch.abs(r - t)
    xy = torch.where(diff < 1.0 / sigma_sq, 0.5 * sigma_sq * diff ** 2, diff - 0.5 / sigma_sq)
    if r.shape[0] == 0:
        orth = r.sum(dim=1)
    elif kink_ref is None:
        orth = torch.mean(torch.abs(_orth_features(r) - _orth_features(t)), dim=1)
    else:
        ft = _orth_features(t)
        sgn = torch.sign(_orth_features(_t(kink_ref, y_pred.dtype)[pos]) - ft).detach()
        orth = torch.mean(sgn * (_orth_features(r) - ft), dim=1)
    normalizer = float(max(1, int(pos.sum())))
    return weight * (0.8 * xy.sum() / normalizer + 0.2 * orth.sum() / normalizer)


REG_L2 = 0.001  # models/retinanet.py:108, all five regression-head kernels
REG_L2_LAYERS = ["reg_conv0", "reg_conv1", "reg_conv2", "reg_conv3", "reg_out"]


def frozen_layer(name):
    """models/resnet.py:100-103: layers[i < 40] that are not BN -> conv1 and every res2* conv."""
    return name == "conv1" or name.startswith("res2")


def trainable_names(W):
    names = []
    for k in W:
        layer, kind = k.split("/")
        if kind not in ("kernel", "bias"):
            continue  # frozen BN
        if frozen_layer(layer):
            continue
        names.append(k)
    return sorted(names)


def loss_and_grads(W, x_nhwc, y_box, y_cls, y_mask, num_classes, dtype=torch.float64, blocks=None, pyramid="sparse",
                   relu_masks=None, loss_params=None, box_kink_ref=None):
    """Total Keras training loss = orthogonal_l1('3Dbox') + focal('cls') + focal('mask') + L2 reg
    (bin/train.py:95-102), and its gradient w.r.t. every trainable tensor.
    relu_masks: see _relu (tests).  box_kink_ref: see orthogonal_l1 (tests).  loss_params: dict(box=(weight, sigma), cls=(alpha, gamma), mask=(alpha, gamma)),
    default = what bin/train.py:97-99 compiles."""
    lp = dict(box=(0.125, 3.0), cls=(0.25, 2.0), mask=(0.25, 2.0))
    lp.update(loss_params or {})
    names = trainable_names(W)
    Wt = {k: _t(v, dtype) for k, v in W.items()}
    for k in names:
        Wt[k].requires_grad_(True)
    out = forward(Wt, x_nhwc, num_classes, dtype, blocks, pyramid=pyramid, relu_masks=relu_masks)
    l_box = orthogonal_l1(_t(y_box, dtype), out["3Dbox"], *lp["box"], kink_ref=box_kink_ref)
    l_cls = focal(_t(y_cls, dtype), out["cls"], *lp["cls"])
    l_mask = focal(_t(y_mask, dtype), out["mask"], *lp["mask"])
    l_reg = sum(REG_L2 * (Wt[n + "/kernel"] ** 2).sum() for n in REG_L2_LAYERS)
    total = l_box + l_cls + l_mask + l_reg
    grads = torch.autograd.grad(total, [Wt[k] for k in names], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(Wt[k])) for k, gr in zip(names, grads)}
    losses = {"3Dbox": float(l_box.detach()), "cls": float(l_cls.detach()), "mask": float(l_mask.detach()),
              "l2": float(l_reg.detach()), "total": float(total.detach())}
    return losses, g, out


def adam_clipnorm_step(W, grads, m, v, step, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-7, clipnorm=0.001):
    """keras 2.3.1 optimizers.Adam.get_updates with Optimizer.get_gradients' global-norm clipnorm:
    norm = sqrt(sum_g sum(g^2)); if norm >= clipnorm: g *= clipnorm / norm  (clip_norm)."""
    names = sorted(grads)
    norm = math.sqrt(sum(float((grads[k].double() ** 2).sum()) for k in names))
    scale = clipnorm / norm if (clipnorm > 0 and norm >= clipnorm) else 1.0
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    new_w = {}
    for k in names:
        g = grads[k] * scale
        m[k] = beta1 * m[k] + (1 - beta1) * g
        v[k] = beta2 * v[k] + (1 - beta2) * g * g
        new_w[k] = torch.as_tensor(np.asarray(W[k]), dtype=g.dtype) - lr_t * m[k] / (torch.sqrt(v[k]) + eps)
    return new_w, norm
