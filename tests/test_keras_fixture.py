"""The Keras pin, when someone supplies it.  tensorflow / keras cannot run in this image, so the graph oracle
(oracle/model_torch.py) is "parity unpinned" (DESIGN.md §1).  Drop the triple written by tools/keras_dump_fixture.py into
tests/golden/keras/ -- weights.npz, x.npy, outputs.npz -- and these tests check (CPU) the oracle and (GPU) the HIP engine
against the REAL Keras outputs on the same weights and input: head outputs within 1e-3 per row (tests/parity_util.py).
Without the files they skip and say so.  The name mapping that produces weights.npz is tested below on a synthetic file
image (always runs)."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keras")
FILES = [os.path.join(HERE, n) for n in ("weights.npz", "x.npy", "outputs.npz")]
HAVE = all(os.path.exists(f) for f in FILES)
WHY = "no Keras fixture: run tools/keras_dump_fixture.py where tensorflow 2.1 / keras 2.3.1 exist and copy its output to tests/golden/keras/"


def _load():
    W = dict(np.load(FILES[0]))
    x = np.load(FILES[1])
    out = dict(np.load(FILES[2]))
    return W, x, out


@pytest.mark.skipif(not HAVE, reason=WHY)
def test_oracle_matches_keras_outputs():
    import torch
    from oracle import model_torch as MT
    from tests.parity_util import assert_rows_within
    W, x, out = _load()
    C = out["cls"].shape[-1]
    with torch.no_grad():
        ref = MT.forward(W, x, C, torch.float64)
    for k in ("3Dbox", "cls", "mask"):
        assert_rows_within(ref[k].numpy(), out[k], k, 1e-3)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason=WHY)
def test_engine_matches_keras_outputs():
    import torch
    from pyrapose_amd.engine import Engine
    from pyrapose_amd.runtime import default_context
    from tests.parity_util import assert_rows_within
    W, x, out = _load()
    C = out["cls"].shape[-1]
    B, H, Wd, _ = x.shape
    eng = Engine(default_context(), C, B, H, Wd, weights=W, train=False)
    eng.forward(torch.from_numpy(x).cuda())
    reg, cls, mask = [t.cpu().numpy() for t in eng.export_outputs()]
    for k, got in (("3Dbox", reg), ("cls", cls), ("mask", mask)):
        assert_rows_within(got, out[k], k, 1e-3)


def test_keras_name_mapping_round_trip():
    """A synthetic Keras file image (auto-named conv2d_<k> with an arbitrary starting k, sub-model groups, BN statistics
    under their Keras names) maps back to exactly the tensors it was made from."""
    from pyrapose_amd import arch
    from pyrapose_amd.utils import keras_names as KN
    C = 5
    W = arch.init_weights(C, seed=3)
    layers = KN.tensors_to_keras(W, first_auto_index=37, reg_model="model_4", cls_model="model_5")
    assert "mask" in layers and "model_4" in layers and "P3" in layers and "bn_conv1" in layers
    assert "conv2d_37/kernel:0" in layers["model_4"] and "bn_conv1/moving_variance:0" in layers["bn_conv1"]
    back = KN.keras_to_tensors(layers, expected_shapes={k: v.shape for k, v in W.items()})
    assert sorted(back) == sorted(W)
    for k in W:
        assert np.array_equal(back[k], np.asarray(W[k], np.float32)), k
    # a file with a missing head conv is refused, not silently mis-dealt
    broken = {g: dict(ws) for g, ws in layers.items()}
    del broken["model_5"]["conv2d_44/kernel:0"], broken["model_5"]["conv2d_44/bias:0"]
    with pytest.raises(ValueError):
        KN.keras_to_tensors(broken)
