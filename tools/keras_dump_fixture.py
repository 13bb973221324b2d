#!/usr/bin/env python3
"""Make the Keras pin: run on a machine with tensorflow 2.1 / keras 2.3.1 / keras-resnet 0.1.0 and the reference checkout
on PYTHONPATH.  Builds the reference training model (random init, seeded), saves its weights, runs it on a seeded
640x480 input and writes the triple that tests/test_keras_fixture.py ingests:
    tests/golden/keras/weights.npz   (this package's tensor names: tools/h5_to_npz.py mapping)
    tests/golden/keras/x.npy         (1, 480, 640, 3) float32, caffe-preprocessed
    tests/golden/keras/outputs.npz   '3Dbox' (1, 56700, 16), 'cls' (1, 56700, C), 'mask' (1, 4800, C)
Nothing of the reference's source is copied: only tensors leave that machine.
usage: python tools/keras_dump_fixture.py [--classes 13] [--out tests/golden/keras]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", type=int, default=13)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "keras"))
    args = ap.parse_args()
    import tensorflow as tf
    from PyraPose import models  # the reference package
    from tools.h5_to_npz import read_keras_h5
    from pyrapose_amd.utils import keras_names as KN
    tf.random.set_seed(0)
    np.random.seed(0)
    model = models.backbone("resnet50").retinanet(args.classes)
    os.makedirs(args.out, exist_ok=True)
    h5 = os.path.join(args.out, "weights.h5")
    model.save_weights(h5)
    W = KN.keras_to_tensors(read_keras_h5(h5))
    np.savez(os.path.join(args.out, "weights.npz"), **W)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, size=(1, 480, 640, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    np.save(os.path.join(args.out, "x.npy"), x)
    outs = model.predict_on_batch(x)
    np.savez(os.path.join(args.out, "outputs.npz"), **{n: np.asarray(o) for n, o in zip(model.output_names, outs)})
    os.remove(h5)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
