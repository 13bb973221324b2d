"""Mirror of ``retinanet_bbox`` (models/retinanet.py:302-335)."""
from . import assert_training_model
from .model import PredictionModel


def retinanet_bbox(model=None, nms=True, class_specific_filter=True, name="retinanet-bbox", anchor_params=None, **kwargs):
    # like the reference graph, `nms` and `class_specific_filter` are accepted and ignored:
    # the prediction model returns raw [boxes3D, classification, mask]
    if model is None:
        raise ValueError("retinanet_bbox: pass the training model")
    assert_training_model(model)
    return PredictionModel(model, anchor_params=anchor_params, name=name)
