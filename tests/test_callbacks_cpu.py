"""CPU: the callbacks of bin/train.py:107-155 -- ModelCheckpoint under the reference's snapshot names, RedirectModel
(callbacks/common.py), ReduceLROnPlateau -- and the name-agnostic weight container (no GPU needed: no engine is built)."""
import os

import numpy as np
import pytest

from pyrapose_amd import callbacks, models, optimizers


def test_model_checkpoint_redirect_and_reload(tmp_path):
    model = models.backbone("resnet50").retinanet(num_classes=4)
    other = models.backbone("resnet50").retinanet(num_classes=4)  # stands in for the wrapped / parallel training model
    name = os.path.join(str(tmp_path), "{backbone}_{dataset_type}_{{epoch:02d}}.h5".format(backbone="resnet50", dataset_type="linemod"))
    cp = callbacks.RedirectModel(callbacks.ModelCheckpoint(name), model)  # bin/train.py:131-143
    cp.set_model(other)
    cp.on_train_begin()
    assert cp.callback.model is model  # redirected
    cp.on_epoch_end(0, {"loss": 1.0})
    cp.on_epoch_end(1, {"loss": 0.5})
    files = sorted(os.listdir(str(tmp_path)))
    assert files == ["resnet50_linemod_01.h5", "resnet50_linemod_02.h5"]  # exactly the reference's names, no temporary left behind
    for f in files:  # '.h5' snapshots ARE HDF5 files in Keras' save_weights layout (what a Keras user can open)
        with open(os.path.join(str(tmp_path), f), "rb") as fh:
            assert fh.read(8) == b"\x89HDF\r\n\x1a\n"
    from pyrapose_amd.utils import hdf5_lite
    f0 = hdf5_lite.File(os.path.join(str(tmp_path), files[0]))  # Keras' model.save layout: the weights sit in 'model_weights'
    assert "layer_names" in f0["model_weights"].attrs and "training_config" in f0.attrs
    wo = callbacks.ModelCheckpoint(os.path.join(str(tmp_path), "w.h5"), save_weights_only=True)
    wo.set_model(model)
    wo.on_epoch_end(0, {})
    assert "layer_names" in hdf5_lite.File(os.path.join(str(tmp_path), "w.h5")).attrs  # save_weights layout: at the root
    os.remove(os.path.join(str(tmp_path), "w.h5"))
    loaded = models.load_model(os.path.join(str(tmp_path), files[1]), backbone_name="resnet50")
    w0, w1 = model.get_weights_dict(), loaded.get_weights_dict()
    assert set(w0) == set(w1) and all(np.array_equal(w0[k], w1[k]) for k in w0)
    assert w0["reg_conv0/kernel"].shape == (3, 3, 256, 512)  # Keras HWIO


@pytest.mark.parametrize("name", ["full.h5", "full.npz"])
def test_full_model_snapshot_round_trips_optimizer_state_and_compile_state(tmp_path, name):
    """`model.save` carries Adam's iterations / m / v and the compile state; `models.load_model` restores
    them (bin/train.py:131-142 writes full models, :336-343 resumes from them).  No GPU: the state waits for the first training plan."""
    import json
    from pyrapose_amd import losses
    from pyrapose_amd.utils import hdf5_lite, keras_names
    model = models.backbone("resnet50").retinanet(num_classes=3)
    model.compile(loss={"3Dbox": losses.orthogonal_l1(weight=0.25, sigma=2.0), "cls": losses.focal(alpha=0.3), "mask": losses.focal(gamma=1.5)},
                  optimizer=optimizers.Adam(lr=3e-5, clipnorm=0.01, beta_2=0.99))
    assert model.optimizer_state() is None
    order = keras_names.trainable_tensor_order(model)
    assert order[0] == "res3a_branch2a/kernel" and "conv1/kernel" not in order and "reg_out/bias" in order and "res3a_branch2a/bias" not in order
    W = model.get_weights_dict()
    rng = np.random.default_rng(0)
    state = dict(iterations=1234, m={k: rng.standard_normal(W[k].shape).astype(np.float32) for k in order},
                 v={k: rng.random(W[k].shape).astype(np.float32) for k in order})
    model.set_optimizer_state(state)
    path = os.path.join(str(tmp_path), name)
    model.save(path)
    if name.endswith(".h5"):
        ow, attrs = hdf5_lite.read_optimizer_weights(path)
        names = list(ow)
        n = len(order)
        assert names[0] == "Adam/iterations:0" and ow[names[0]].dtype == np.int64 and int(ow[names[0]]) == 1234
        assert names[1] == "training/Adam/m_0:0" and names[1 + n] == "training/Adam/v_0:0" and names[1 + 2 * n] == "training/Adam/vhat_0:0"
        assert len(names) == 1 + 3 * n and ow[names[-1]].shape == (1,)
        tc = json.loads(bytes(attrs["training_config"]).decode())
        assert tc["optimizer_config"]["class_name"] == "Adam" and tc["optimizer_config"]["config"]["clipnorm"] == 0.01
        assert tc["loss"] == {"3Dbox": "orthogonal_l1", "cls": "focal", "mask": "focal"}
    back = models.load_model(path, backbone_name="resnet50")
    assert back.num_classes == 3 and back._optimizer.lr == 3e-5 and back._optimizer.beta_2 == 0.99 and back._optimizer.clipnorm == 0.01
    assert back._loss["3Dbox"].weight == 0.25 and back._loss["3Dbox"].sigma == 2.0 and back._loss["cls"].alpha == 0.3 and back._loss["mask"].gamma == 1.5
    got = back.optimizer_state()
    assert got["iterations"] == 1234 and list(got["m"]) == order
    assert all(np.array_equal(got["m"][k], state["m"][k]) and np.array_equal(got["v"][k], state["v"][k]) for k in order)
    w1 = back.get_weights_dict()
    assert all(np.array_equal(W[k], w1[k]) for k in W)
    # weights-only loading of a full snapshot ignores the optimizer part; recompiling drops the restored state, like Keras
    m2 = models.backbone("resnet50").retinanet(num_classes=3)
    m2.load_weights(path)
    assert np.array_equal(m2.get_weights_dict()["P5/kernel"], W["P5/kernel"]) and m2.optimizer_state() is None
    back.compile(loss=back._loss, optimizer=optimizers.Adam(lr=1e-5, clipnorm=0.001))
    assert back.optimizer_state() is None


def test_checkpoint_container_escape_and_formats(tmp_path, monkeypatch):
    model = models.backbone("resnet50").retinanet(num_classes=2)
    a = os.path.join(str(tmp_path), "a.h5")
    monkeypatch.setenv("PP_CHECKPOINT_NPZ", "1")                 # the escape: numpy container under the '.h5' name
    model.save(a)
    with open(a, "rb") as fh:
        assert fh.read(2) == b"PK"
    monkeypatch.delenv("PP_CHECKPOINT_NPZ")
    b = os.path.join(str(tmp_path), "b.npz")
    model.save_weights(b)                                        # other names: the container
    with open(b, "rb") as fh:
        assert fh.read(2) == b"PK"
    with pytest.raises(ValueError):
        model.save_weights(os.path.join(str(tmp_path), "c.h5"), format="hdf")
    for path in (a, b):
        m = models.backbone("resnet50").retinanet(num_classes=2)
        m._weights = {k: np.zeros_like(v) for k, v in m.get_weights_dict().items()}
        m.load_weights(path)
        assert np.array_equal(m.get_weights_dict()["P4/kernel"], model.get_weights_dict()["P4/kernel"])
    assert sorted(os.listdir(str(tmp_path))) == ["a.h5", "b.npz"]


def test_checkpoint_save_best_only_and_period(tmp_path):
    model = models.backbone("resnet50").retinanet(num_classes=2)
    cp = callbacks.ModelCheckpoint(os.path.join(str(tmp_path), "best.h5"), monitor="loss", save_best_only=True)
    cp.set_model(model)
    for epoch, loss in enumerate([3.0, 2.0, 2.5, 1.0]):
        before = os.path.getmtime(cp.filepath) if os.path.exists(cp.filepath) else None
        cp.on_epoch_end(epoch, {"loss": loss})
        if loss == 2.5:
            assert os.path.getmtime(cp.filepath) == before  # no improvement, no write
    assert cp.best == 1.0
    every2 = callbacks.ModelCheckpoint(os.path.join(str(tmp_path), "e{epoch:02d}.h5"), period=2)
    every2.set_model(model)
    for epoch in range(4):
        every2.on_epoch_end(epoch, {})
    assert sorted(f for f in os.listdir(str(tmp_path)) if f.startswith("e")) == ["e02.h5", "e04.h5"]


def test_hdf5_is_recognised_by_content_not_by_name(tmp_path):
    """a file that starts with the HDF5 signature goes to the subset reader whatever it is called: a broken one is refused
    with the conversion command, a Keras-layout one (here from our own writer) loads"""
    import numpy as np
    model = models.backbone("resnet50").retinanet(num_classes=2)
    p = os.path.join(str(tmp_path), "keras.h5")
    with open(p, "wb") as f:
        f.write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(ImportError) as e:
        model.load_weights(p)
    assert "h5_to_npz" in str(e.value)
    good = os.path.join(str(tmp_path), "snapshot.weights")
    model.save_weights(good, format="h5")
    with open(good, "rb") as f:
        assert f.read(4) == b"\x89HDF"
    m4 = models.load_model(good, backbone_name="resnet50")      # class count read from the file
    assert m4.num_classes == 2 and np.array_equal(m4.get_weights_dict()["P3/kernel"], model.get_weights_dict()["P3/kernel"])
    m2 = models.backbone("resnet50").retinanet(num_classes=2)
    m2._weights = {k: np.zeros_like(v) for k, v in m2.get_weights_dict().items()}
    m2.load_weights(good)
    a, b = model.get_weights_dict(), m2.get_weights_dict()
    assert sorted(a) == sorted(b) and all(np.array_equal(a[k], b[k]) for k in a)
    # by_name on a backbone-only file (the ImageNet weights of models/resnet.py:89-98): named layers load, heads stay
    from pyrapose_amd.utils import hdf5_lite, keras_names
    layers = keras_names.tensors_to_keras(a)
    part = {g: w for g, w in layers.items() if g.startswith(("conv1", "bn", "res"))}
    hdf5_lite.write_keras_weights(os.path.join(str(tmp_path), "imagenet.h5"), part)
    m3 = models.backbone("resnet50").retinanet(num_classes=2)
    before = m3.get_weights_dict()["reg_out/kernel"].copy()
    m3._weights = dict(m3.get_weights_dict(), **{"res3a_branch2a/kernel": np.zeros_like(a["res3a_branch2a/kernel"])})
    m3.load_weights(os.path.join(str(tmp_path), "imagenet.h5"), by_name=True, skip_mismatch=True)
    c = m3.get_weights_dict()
    assert np.array_equal(c["res3a_branch2a/kernel"], a["res3a_branch2a/kernel"]) and np.array_equal(c["reg_out/kernel"], before)


def test_reduce_lr_on_plateau_schedule():
    model = models.backbone("resnet50").retinanet(num_classes=2)
    model.compile(loss={"3Dbox": __import__("pyrapose_amd").losses.orthogonal_l1(), "cls": __import__("pyrapose_amd").losses.focal(),
                        "mask": __import__("pyrapose_amd").losses.focal()}, optimizer=optimizers.Adam(lr=1e-4, clipnorm=0.001))
    cb = callbacks.ReduceLROnPlateau(monitor="loss", factor=0.1, patience=2, verbose=0, min_delta=0.0001)  # bin/train.py:145-154
    cb.set_model(model)
    for epoch, loss in enumerate([1.0, 0.9, 0.9, 0.9, 0.9]):
        cb.on_epoch_end(epoch, {"loss": loss})
    assert model.lr == pytest.approx(1e-5)
