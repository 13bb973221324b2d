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
    """models/__init__.py:68-71.  ``filepath`` is an .npz written by ``model.save``; the class count is read from it."""
    import numpy as np
    from .model import PyraPoseModel
    with open(filepath, "rb") as f:
        is_hdf5 = f.read(4) == b"\x89HDF"
    if is_hdf5:  # a real Keras file: the subset reader + name mapping (utils/hdf5_lite.py, utils/keras_names.py)
        from ..utils import hdf5_lite, keras_names
        data = keras_names.keras_to_tensors(hdf5_lite.read_keras_weights(filepath))
    else:
        data = np.load(filepath)
    if num_classes is None:
        num_classes = int(data["mask_out/bias"].shape[0])
    m = PyraPoseModel(num_classes, backbone_name.split("_")[0])
    m.load_weights(filepath)
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
