"""Mirror of the reference's ``PyraPose/models/resnet.py`` (:29-110)."""
import numpy as np

from . import Backbone
from . import retinanet as _retinanet  # noqa: F401
from .model import PyraPoseModel


def preprocess_image(x, mode="caffe"):
    """utils/image.py:35-62 'caffe' mode: float32, subtract the ImageNet BGR means (input is BGR)."""
    x = np.asarray(x).astype(np.float32)
    if mode == "caffe":
        x[..., 0] -= 103.939
        x[..., 1] -= 116.779
        x[..., 2] -= 123.68
    elif mode == "tf":
        x /= 127.5
        x -= 1.0
    return x


class ResNetBackbone(Backbone):
    def retinanet(self, *args, **kwargs):
        return resnet_retinanet(*args, backbone=self.backbone, **kwargs)

    def download_imagenet(self):
        raise IOError("download_imagenet: no network in this environment; pass --weights <file.npz> instead "
                      "(the reference fetches ResNet-50-model.keras.h5 from fizyr/keras-models, models/resnet.py:42-62)")

    def validate(self):
        allowed_backbones = ["resnet50", "resnet101", "resnet152"]
        backbone = self.backbone.split("_")[0]
        if backbone not in allowed_backbones:
            raise ValueError("Backbone ('{}') not in allowed backbones ({}).".format(backbone, allowed_backbones))

    def preprocess_image(self, inputs):
        return preprocess_image(inputs, mode="caffe")


def resnet_retinanet(num_classes, inputs=None, modifier=None, backbone="resnet50", num_anchors=None, **kwargs):
    """models/resnet.py:79-110.  The reference hard-codes ResNet50 whatever the name says; we honour
    the requested depth.  ``modifier`` (freeze_model, bin/train.py:74) marks the backbone frozen."""
    # create_pyramid_features (models/retinanet.py:265) is selected by name here: pyramid='sparse' | 'fpn' | 'p3p7'
    m = PyraPoseModel(num_classes, backbone.split("_")[0], freeze_backbone=modifier is not None,
                      pyramid=kwargs.get("pyramid", "sparse"), anchor_params=kwargs.get("anchor_params"))
    if num_anchors is not None and num_anchors != m.anchor_params.num_anchors():
        raise ValueError("num_anchors=%d does not match anchor_params (%d)" % (num_anchors, m.anchor_params.num_anchors()))
    return m


def resnet50_retinanet(num_classes, inputs=None, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone="resnet50", inputs=inputs, **kwargs)
