"""Keras-2.3.1 weight names of the reference graph  <->  the tensor names of this package.

The reference names the ResNet layers explicitly (keras_resnet: conv1, bn_conv1, res2a_branch2a, bn2a_branch2a, ...) and the
three output convs of its feature pyramid (models/retinanet.py:204,209,212: 'P3', 'P4', 'P5'); every other Conv2D is unnamed
and gets Keras' auto name conv2d_<k>, k counting Conv2D creations of the process.  retinanet() (models/retinanet.py:260-299)
creates them in this order: the regression sub-model (4 trunk convs + output, :101-131), the classification sub-model
(:9-54), the mask sub-model (:57-98, Model name 'mask'), then __create_sparceFPN (:180-214): lateral 1x1 on C3, C4, C5, the
3x3 on the level-4 sum, the 3x3 on the level-3 sum, the stride-2 3x3 on level 3, the stride-2 3x3 on level 4.  A saved file
stores the sub-models' weights under their Model layer ('model_<j>', 'model_<j+1>', 'mask') with weight names
'conv2d_<k>/kernel:0'.  So: sort the auto-named convs by k and deal them out in that order; shapes are checked.

keras_to_tensors() is pure (dict in, dict out): tools/h5_to_npz.py feeds it from h5py on a machine that has it, the tests
feed it a synthetic file image."""
import re
from collections import OrderedDict

import numpy as np

AUTO_ORDER = (["reg_conv%d" % i for i in range(4)] + ["reg_out"] + ["cls_conv%d" % i for i in range(4)] + ["cls_out"] +
              ["mask_conv%d" % i for i in range(4)] + ["mask_out"] +
              ["fpn_lat3", "fpn_lat4", "fpn_lat5", "fpn_mid4", "fpn_mid3", "fpn_down3", "fpn_down4"])
_BN = {"gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}


def _strip(name):
    name = name.decode() if isinstance(name, bytes) else str(name)
    return name[:-2] if name.endswith(":0") else name


def keras_to_tensors(layers, expected_shapes=None, partial=False):
    """layers: {layer_group_name: {weight_name: array}} as read from a Keras .h5 ('model_weights' group or file root;
    weight names like 'res2a_branch2a/kernel:0', 'bn2a_branch2a/moving_mean:0', 'conv2d_7/bias:0').
    Returns {'<layer>/kernel' (HWIO), '<layer>/bias', '<bn>/{gamma,beta,mean,var}'} with this package's layer names.
    expected_shapes: optional {name: shape} (e.g. from arch.init_weights) checked against the result.
    partial: the file holds only part of the graph (the ImageNet backbone of models/resnet.py:89-98, loaded by_name): map the
    explicitly named layers, deal out the auto-named convs only when all of them are there, check shapes of what was found."""
    flat = OrderedDict()
    for group, ws in layers.items():
        for wname, arr in ws.items():
            flat[_strip(wname)] = np.asarray(arr)
    out, auto = OrderedDict(), {}
    for full, arr in flat.items():
        layer, _, var = full.rpartition("/")
        layer = layer.split("/")[-1]  # weight names may repeat the enclosing model's name
        m = re.fullmatch(r"conv2d(?:_(\d+))?", layer)
        if m:
            auto.setdefault(int(m.group(1) or 0), {})[var] = arr
        elif var in _BN:
            out["%s/%s" % (layer, _BN[var])] = arr.astype(np.float32)
        elif var in ("kernel", "bias"):
            out["%s/%s" % (layer, var)] = arr.astype(np.float32)
        else:
            raise ValueError("unexpected Keras weight %r" % full)
    ks = sorted(auto)
    if partial and len(ks) != len(AUTO_ORDER):
        ks = []
    elif len(ks) != len(AUTO_ORDER):
        raise ValueError("expected %d auto-named Conv2D layers (heads + FPN), found %d: %s" % (len(AUTO_ORDER), len(ks), ks))
    for k, ours in zip(ks, AUTO_ORDER):
        for var, arr in auto[k].items():
            out["%s/%s" % (ours, var)] = arr.astype(np.float32)
    if expected_shapes is not None:
        for name, shape in expected_shapes.items():
            if name not in out:
                if partial:
                    continue
                raise ValueError("Keras file has no tensor for %s" % name)
            if tuple(out[name].shape) != tuple(shape):
                raise ValueError("shape of %s: Keras %s, expected %s" % (name, out[name].shape, tuple(shape)))
    return out


def tensors_to_keras(W, first_auto_index=1, reg_model="model_1", cls_model="model_2"):
    """The inverse (for tests and for exporting a checkpoint that Keras' load_weights(by_name) would accept):
    {layer_group: {weight_name: array}} in the layout keras_to_tensors reads."""
    inv_bn = {v: k for k, v in _BN.items()}
    auto_of = {n: first_auto_index + i for i, n in enumerate(AUTO_ORDER)}
    layers = OrderedDict()
    for key, arr in W.items():
        layer, var = key.split("/")
        if layer in auto_of:
            kname = "conv2d_%d" % auto_of[layer]
            group = reg_model if layer.startswith("reg_") else cls_model if layer.startswith("cls_") else "mask" if layer.startswith("mask_") else kname
            layers.setdefault(group, OrderedDict())["%s/%s:0" % (kname, var)] = np.asarray(arr)
        else:
            v = inv_bn.get(var, var)
            layers.setdefault(layer, OrderedDict())["%s/%s:0" % (layer, v)] = np.asarray(arr)
    return layers


def trainable_tensor_order(model):
    """'<layer>/kernel', '<layer>/bias' of the tensors Adam updates, in this package's graph order (= the order in which
    PyraPoseModel.save numbers Keras' m_<i> / v_<i>): conv1 / res2 and every BatchNormalization are frozen (models/resnet.py:87-103)."""
    from .. import arch
    out = []
    for s in arch.all_specs(model.num_classes, model.backbone_name, model.pyramid, model.anchor_params.num_anchors()):
        if s.trainable and not (model.freeze_backbone and s.bn):
            out.append(s.name + "/kernel")
            if s.bias:
                out.append(s.name + "/bias")
    return out
