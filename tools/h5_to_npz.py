#!/usr/bin/env python3
"""Convert a Keras-2.3.1 HDF5 file of the reference model (model.save / save_weights: bin/train.py:107-128 checkpoints,
models/resnet.py:89-98 ImageNet file) into the .npz this package loads (PyraPoseModel.load_weights).  Needs h5py -- run it
where the Keras file was made; this image has none.  The name mapping itself lives in
pyrapose_amd/utils/keras_names.py and is unit-tested on a synthetic file image.

usage: python tools/h5_to_npz.py model.h5 out.npz [--partial]     (--partial: backbone-only files, e.g. the ImageNet weights)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_keras_h5(path):
    import h5py  # not in this image: run on the machine that holds the Keras file
    layers = {}
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        for lname in g.attrs["layer_names"]:
            lname = lname.decode() if isinstance(lname, bytes) else lname
            grp = g[lname]
            names = [n.decode() if isinstance(n, bytes) else n for n in grp.attrs["weight_names"]]
            if names:
                layers[lname] = {n: np.asarray(grp[n]) for n in names}
    return layers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("h5")
    ap.add_argument("npz")
    ap.add_argument("--partial", action="store_true", help="the file holds only some layers (ImageNet backbone): map what is there")
    args = ap.parse_args()
    from pyrapose_amd.utils import keras_names as KN
    layers = read_keras_h5(args.h5)
    if args.partial:
        layers = {k: v for k, v in layers.items() if not any(w.split("/")[-2].startswith("conv2d") for w in v)}
        W = {}
        for grp in layers.values():
            for wname, arr in grp.items():
                layer, var = KN._strip(wname).split("/")[-2:]
                W["%s/%s" % (layer, KN._BN.get(var, var))] = np.asarray(arr, np.float32)
    else:
        W = KN.keras_to_tensors(layers)
    np.savez(args.npz, **W)
    print("wrote %s: %d tensors" % (args.npz, len(W)))


if __name__ == "__main__":
    main()
