"""Static description of the PyraPose network: layer table, Keras-style initialisers, weight dicts.

Mirrors what the reference builds at models/resnet.py:79-110 (keras_resnet ResNet50, frozen BN,
first 40 non-BN layers frozen) and models/retinanet.py:9-131,180-214 (heads, __create_sparceFPN).
Weight dict keys: '<layer>/kernel' (HWIO, as Keras stores it), '<layer>/bias',
'<bn>/{gamma,beta,mean,var}'.  FPN / head layers are unnamed in the reference (Keras auto-names);
we name them fpn_lat{3,4,5}, fpn_mid{3,4}, fpn_down{3,4}, P3, P4, P5, {reg,cls,mask}_conv{0..3},
{reg,cls,mask}_out.
"""
import math
from collections import OrderedDict

import numpy as np

BACKBONE_BLOCKS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}
BN_EPS = 1e-5
NUM_ANCHORS = 9
REG_L2 = 0.001


class ConvSpec(object):
    def __init__(self, name, cin, cout, k, stride=1, pad="same", bias=True, bn=None, trainable=True, l2=0.0,
                 init="glorot_uniform", bias_init=0.0):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.bias, self.bn, self.trainable, self.l2, self.init, self.bias_init = bias, bn, trainable, l2, init, bias_init


def block_name(stage, block, numerical):
    # keras_resnet: 'a', then 'b', 'c', ... or 'b1', 'b2', ... when numerical_names is set (R-101/152 stages 3,4)
    if block == 0 or not numerical:
        return chr(ord("a") + block)
    return "b%d" % block


def backbone_specs(backbone="resnet50"):
    blocks = BACKBONE_BLOCKS[backbone]
    numerical = {"resnet50": [False] * 4, "resnet101": [False, True, True, False], "resnet152": [False, True, True, False]}[backbone]
    specs = [ConvSpec("conv1", 3, 64, 7, 2, pad=3, bias=False, bn="bn_conv1", trainable=False, init="he_normal")]
    cin = 64
    for stage, n_blocks in enumerate(blocks):
        f = 64 * 2 ** stage
        for block in range(n_blocks):
            sc, bc = str(stage + 2), block_name(stage, block, numerical[stage])
            stride = 1 if (block != 0 or stage == 0) else 2
            trainable = stage != 0  # models/resnet.py:100-103 freezes conv1 + every res2* conv
            mk = lambda br, ci, co, k, s, p: ConvSpec("res%s%s_branch%s" % (sc, bc, br), ci, co, k, s, pad=p, bias=False,
                                                      bn="bn%s%s_branch%s" % (sc, bc, br), trainable=trainable, init="he_normal")
            specs.append(mk("2a", cin, f, 1, stride, 0))
            specs.append(mk("2b", f, f, 3, 1, 1))
            specs.append(mk("2c", f, 4 * f, 1, 1, 0))
            if block == 0:
                specs.append(mk("1", cin, 4 * f, 1, stride, 0))
            cin = 4 * f
    return specs


PYRAMIDS = ("sparse", "fpn", "p3p7")
PYRAMID_LEVELS = {"sparse": (3, 4, 5), "fpn": (3, 4, 5), "p3p7": (3, 4, 5, 6, 7)}


def fpn_specs(pyramid="sparse"):
    """'sparse' = __create_sparceFPN (models/retinanet.py:180-214, what retinanet() builds on master);
    'fpn' = __create_FPN (:160-177); 'p3p7' = __create_pyramid_features (:134-157).  The lateral 1x1 convs are unnamed
    in the reference (Keras auto-names); here C{3,4,5}_reduced.  P{3..7}_con are the reference's names."""
    if pyramid != "sparse":
        s = [ConvSpec("C5_reduced", 2048, 256, 1), ConvSpec("P5_con", 256, 256, 3), ConvSpec("C4_reduced", 1024, 256, 1),
             ConvSpec("P4_con", 256, 256, 3), ConvSpec("C3_reduced", 512, 256, 1), ConvSpec("P3_con", 256, 256, 3)]
        if pyramid == "p3p7":
            s += [ConvSpec("P6_con", 2048, 256, 3, 2), ConvSpec("P7_con", 256, 256, 3, 2)]
        return s
    s = [ConvSpec("fpn_lat3", 512, 256, 1), ConvSpec("fpn_lat4", 1024, 256, 1), ConvSpec("fpn_lat5", 2048, 256, 1),
         ConvSpec("fpn_mid4", 256, 256, 3), ConvSpec("fpn_mid3", 256, 256, 3), ConvSpec("fpn_down3", 256, 256, 3, 2),
         ConvSpec("P3", 256, 256, 3), ConvSpec("fpn_down4", 256, 256, 3, 2), ConvSpec("P4", 256, 256, 3),
         ConvSpec("P5", 256, 256, 3)]
    return s


def head_specs(num_classes, num_anchors=NUM_ANCHORS):
    prior = -math.log((1 - 0.01) / 0.01)  # initializers.py:23-39 PriorProbability(0.01)
    s = []
    for i in range(4):
        s.append(ConvSpec("reg_conv%d" % i, 256 if i == 0 else 512, 512, 3, l2=REG_L2, init="normal001"))
    s.append(ConvSpec("reg_out", 512, num_anchors * 16, 3, l2=REG_L2, init="normal001"))
    for i in range(4):
        s.append(ConvSpec("cls_conv%d" % i, 256, 256, 3, init="normal001"))
    s.append(ConvSpec("cls_out", 256, num_anchors * num_classes, 3, init="normal001", bias_init=prior))
    for i in range(4):
        s.append(ConvSpec("mask_conv%d" % i, 256, 256, 3, init="normal001"))
    s.append(ConvSpec("mask_out", 256, num_classes, 3, init="normal001", bias_init=prior))
    return s


def all_specs(num_classes, backbone="resnet50", pyramid="sparse", num_anchors=NUM_ANCHORS):
    assert pyramid in PYRAMIDS, pyramid
    return backbone_specs(backbone) + fpn_specs(pyramid) + head_specs(num_classes, num_anchors)


def init_weights(num_classes, seed=0, backbone="resnet50", pyramid="sparse", num_anchors=NUM_ANCHORS):
    """Random-init weights of the reference architecture (no network for the ImageNet file):
    heads N(0, 0.01) / zeros / PriorProbability (retinanet.py:35-43,80-88,106-107), FPN Keras default
    glorot_uniform + zero bias, backbone He-normal stand-in with a random frozen-BN affine."""
    rng = np.random.default_rng(seed)
    W = OrderedDict()
    for s in all_specs(num_classes, backbone, pyramid, num_anchors):
        fan_in, fan_out = s.k * s.k * s.cin, s.k * s.k * s.cout
        shape = (s.k, s.k, s.cin, s.cout)
        if s.init == "normal001":
            k = rng.normal(0.0, 0.01, size=shape)
        elif s.init == "glorot_uniform":
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            k = rng.uniform(-lim, lim, size=shape)
        else:  # he_normal stand-in for the ImageNet weights
            k = rng.normal(0.0, math.sqrt(2.0 / fan_in), size=shape)
            if s.name == "conv1":
                k = k / 64.0  # the stem sees +-128 pixel values; trained BN would normalise them
        W[s.name + "/kernel"] = k.astype(np.float32)
        if s.bias:
            W[s.name + "/bias"] = np.full((s.cout,), s.bias_init, np.float32)
        if s.bn:
            # frozen random BN cannot normalise: damp the residual branch so activations stay O(1) over 16 blocks
            lo, hi = (0.2, 0.4) if s.name.endswith("branch2c") else (0.5, 1.5)
            W[s.bn + "/gamma"] = rng.uniform(lo, hi, size=s.cout).astype(np.float32)
            W[s.bn + "/beta"] = rng.normal(0, 0.1, size=s.cout).astype(np.float32)
            W[s.bn + "/mean"] = rng.normal(0, 0.1, size=s.cout).astype(np.float32)
            W[s.bn + "/var"] = rng.uniform(0.5, 1.5, size=s.cout).astype(np.float32)
    return W


def level_shapes(h, w, levels=(3, 4, 5)):
    """utils/anchors.py:357-369 guess_shapes == the ResNet shape walk for these strides."""
    return [((h + 2 ** l - 1) // 2 ** l, (w + 2 ** l - 1) // 2 ** l) for l in levels]


def conv_flops(num_classes, h, w, backbone="resnet50"):
    """2*MACs per image: (forward, backward = dgrad + wgrad for trainable convs; no dgrad into frozen C2)."""
    fwd = bwd = 0.0
    hh, ww = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    sizes = {}
    cur = (hh, ww)
    specs = backbone_specs(backbone)
    # walk the backbone
    fwd += 2.0 * hh * ww * 49 * 3 * 64
    cur = ((hh + 1) // 2, (ww + 1) // 2)
    first_trainable_input_done = False
    for s in specs[1:]:
        if s.name.endswith("branch2a") or s.name.endswith("branch1"):
            src = cur_in if s.name.endswith("branch1") else cur
            if s.name.endswith("branch2a"):
                cur_in = cur
            out = ((src[0] - 1) // s.stride + 1, (src[1] - 1) // s.stride + 1)
            if s.name.endswith("branch2a"):
                cur = out
        else:
            out = cur
        fl = 2.0 * out[0] * out[1] * s.k * s.k * s.cin * s.cout
        fwd += fl
        if s.trainable:
            bwd += fl  # wgrad
            needs_dgrad = not (s.name.startswith("res3a") and (s.name.endswith("branch2a") or s.name.endswith("branch1")))
            if needs_dgrad:
                bwd += fl
        sizes[s.name] = out
    lv = level_shapes(h, w)
    cells = [a * b for a, b in lv]
    def add(fl):
        nonlocal fwd, bwd
        fwd += fl
        bwd += 2 * fl
    add(2.0 * cells[0] * 512 * 256); add(2.0 * cells[1] * 1024 * 256); add(2.0 * cells[2] * 2048 * 256)
    k9 = 9 * 256 * 256 * 2.0
    add(cells[1] * k9); add(cells[0] * k9)            # mid4, mid3
    add(cells[1] * k9); add(cells[0] * k9)            # down3 (out = level 4), P3
    add(cells[2] * k9); add(cells[1] * k9); add(cells[2] * k9)  # down4, P4, P5
    tot = sum(cells)
    for s in head_specs(num_classes):
        n = cells[0] if s.name.startswith("mask") else tot
        add(2.0 * n * 9 * s.cin * s.cout)
    return fwd, bwd
