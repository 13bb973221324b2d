"""Mirror of the reference's ``PyraPose/models/__init__.py``: ``backbone(name)`` -> Backbone object,
``load_model``, ``convert_model``, ``assert_training_model``, ``check_training_model`` (:6-91)."""
from __future__ import print_function

import sys


class Backbone(object):
    """models/__init__.py:6-52."""

    def __init__(self, backbone):
        from .. import initializers, layers, losses
        self.custom_objects = {
            "UpsampleLike": layers.UpsampleLike,
            "PriorProbability": initializers.PriorProbability,
            "FilterDetections": layers.FilterDetections,
            "Anchors": layers.Anchors,
            "_focal": losses.focal(),
            "_orth_l1": losses.orthogonal_l1(),
            "RegressBoxes3D": layers.RegressBoxes3D(),
        }
        self.backbone = backbone
        self.validate()

    def retinanet(self, *args, **kwargs):
        raise NotImplementedError("retinanet method not implemented.")

    def validate(self):
        raise NotImplementedError("validate method not implemented.")

    def preprocess_image(self, inputs):
        raise NotImplementedError("preprocess_image method not implemented.")


def backbone(backbone_name):
    """models/__init__.py:55-65 (only the resnet family is on the hot path)."""
    if "resnet" in backbone_name:
        from .resnet import ResNetBackbone as b
    else:
        raise NotImplementedError("Backbone class for  '{}' not implemented.".format(backbone_name))
    return b(backbone_name)


def load_model(filepath, backbone_name="resnet50", num_classes=None):
    """models/__init__.py:68-71 (`keras.models.load_model`): the model of a snapshot written by `model.save` -- weights, and when
    the file carries them the compile state (optimizer + losses) and Adam's iterations / moments, so that the next
    `fit_generator` continues the interrupted run (bin/train.py:336-343).  Also takes a weights-only file (Keras .h5 or the
    numpy container): the class count is then read from the mask head's bias and the model comes back uncompiled."""
    import json

    import numpy as np

    from .. import losses, optimizers
    from ..utils import hdf5_lite, keras_names
    from .model import PyraPoseModel
    cfg, state = {}, None
    if hdf5_lite.is_hdf5(filepath):
        data = keras_names.keras_to_tensors(hdf5_lite.read_keras_weights(filepath))
        ow, attrs = hdf5_lite.read_optimizer_weights(filepath)
        if "pyrapose_amd_config" in attrs:
            cfg = json.loads(bytes(np.asarray(attrs["pyrapose_amd_config"]).tobytes()).rstrip(b"\0").decode("utf-8"))
        if ow is not None:
            names = [v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v)
                     for v in np.atleast_1d(attrs.get("pyrapose_amd_optimizer_tensors", []))]
            vals = list(ow.values())
            n = (len(vals) - 1) // 3  # Keras' Adam: [iterations] + ms + vs + vhats
            if names and len(names) != n:
                raise ValueError("load_model: %d optimizer tensors named, %d stored" % (len(names), n))
            if not names:  # a file without the name list: positional, in the order of this package's trainable tensors
                probe = PyraPoseModel(int(cfg.get("num_classes", data["mask_out/bias"].shape[0])), cfg.get("backbone", backbone_name.split("_")[0]),
                                      pyramid=cfg.get("pyramid", "sparse"))
                names = [l for l in keras_names.trainable_tensor_order(probe)]
            state = dict(iterations=int(np.asarray(vals[0]).reshape(-1)[0]), m=dict(zip(names, vals[1:1 + n])), v=dict(zip(names, vals[1 + n:1 + 2 * n])))
    else:
        z = np.load(filepath)
        data = {k: z[k] for k in z.files if not (k.startswith("optimizer/") or k.startswith("config/"))}
        if "config/json" in z.files:
            cfg = json.loads(z["config/json"].tobytes().decode("utf-8"))
        if "optimizer/iterations" in z.files:
            ms = {k[len("optimizer/m/"):]: z[k] for k in z.files if k.startswith("optimizer/m/")}
            vs = {k[len("optimizer/v/"):]: z[k] for k in z.files if k.startswith("optimizer/v/")}
            state = dict(iterations=int(z["optimizer/iterations"]), m=ms, v=vs)
    if num_classes is None:
        num_classes = int(cfg.get("num_classes", data["mask_out/bias"].shape[0]))
    m = PyraPoseModel(num_classes, cfg.get("backbone", backbone_name.split("_")[0]), pyramid=cfg.get("pyramid", "sparse"),
                      freeze_backbone=bool(cfg.get("freeze_backbone", False)))
    m.load_weights(filepath)
    if "optimizer" in cfg and "loss" in cfg:
        mk = {"orthogonal_l1": lambda d: losses.orthogonal_l1(weight=d.get("weight", 0.125), sigma=d.get("sigma", 3.0)),
              "focal": lambda d: losses.focal(alpha=d.get("alpha", 0.25), gamma=d.get("gamma", 2.0))}
        m.compile(loss={k: mk[d["kind"]](d) for k, d in cfg["loss"].items()}, optimizer=optimizers.Adam(**cfg["optimizer"]["config"]))
    if state is not None:
        m.set_optimizer_state(state)
    return m


def convert_model(model, nms=True, class_specific_filter=True, anchor_params=None):
    """models/__init__.py:74-76."""
    from .retinanet import retinanet_bbox
    return retinanet_bbox(model=model, nms=nms, class_specific_filter=class_specific_filter, anchor_params=anchor_params)


def assert_training_model(model):
    assert all(output in model.output_names for output in ["3Dbox", "cls", "mask"]), \
        "Input is not a training model. Outputs were found, outputs are: {}).".format(model.output_names)


def check_training_model(model):
    try:
        assert_training_model(model)
    except AssertionError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
