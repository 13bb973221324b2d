"""Keras-``Model``-shaped handles over the HIP engine, so that the reference's callers keep working:
bin/train.py:65-68,95-102,361,381-390 (load_weights / compile / summary / fit_generator),
utils/linemod_eval.py:303 (predict_on_batch), models/__init__.py:79-80 (output_names),
models/retinanet.py:323 (get_layer('P3').output).

The Keras graph is shape-polymorphic (Input(None, None, 3)); the engine plans launches for one
(batch, height, width), so a handle builds its engine on first use and rebuilds it (weights carried
over) when the batch shape changes.
"""
import os
from collections import OrderedDict
from struct import error as struct_error

import numpy as np
import torch

from .. import arch, optimizers
from ..engine import Engine
from ..runtime import default_context

OUTPUT_NAMES = ["3Dbox", "cls", "mask"]


class _LayerHandle(object):
    def __init__(self, name, model):
        self.name, self._model = name, model

    @property
    def output(self):
        return ("tensor", self.name)

    def get_weights(self):
        W = self._model.get_weights_dict()
        return [W[k] for k in (self.name + "/kernel", self.name + "/bias") if k in W]


class PyraPoseModel(object):
    """Training model: outputs ['3Dbox' (B,N,16), 'cls' (B,N,C), 'mask' (B,HW/64,C)]."""

    def __init__(self, num_classes, backbone="resnet50", weights=None, seed=0, freeze_backbone=False, name="retinanet",
                 pyramid="sparse", anchor_params=None):
        """pyramid: which `create_pyramid_features` of models/retinanet.py:260-299 to build -- 'sparse' (the default
        there, __create_sparceFPN), 'fpn' (__create_FPN) or 'p3p7' (__create_pyramid_features)."""
        from ..utils.anchors import AnchorParameters
        self.name = name
        self.num_classes = int(num_classes)
        self.backbone_name = backbone
        self.pyramid = pyramid
        self.anchor_params = anchor_params or (AnchorParameters.p3p7 if pyramid == "p3p7" else AnchorParameters.default)
        self.output_names = list(OUTPUT_NAMES)
        na = self.anchor_params.num_anchors()
        self._weights = weights if weights is not None else arch.init_weights(self.num_classes, seed, backbone, pyramid, na)
        self._engine = None          # the plan that holds the freshest weights (the last one used)
        self._engines = OrderedDict()  # (B, H, W, train) -> Engine
        self._weights_version = 0    # bumped by every optimisation step / load_weights
        self._loss = None
        self._optimizer = None
        self._pending_opt = None     # optimizer state read from a snapshot, waiting for the first training plan
        self.freeze_backbone = freeze_backbone
        self.stop_training = False
        self.layers = [_LayerHandle(s.name, self) for s in arch.all_specs(self.num_classes, backbone, pyramid, na)]

    # ---- engine management ---------------------------------------------------------------------
    MAX_ENGINES = 4  # plans kept alive (one per (batch, height, width, train) key), least recently used first out

    def _loss_params(self):
        l = self._loss or {}
        out = {}
        if "3Dbox" in l:
            out["box"] = (l["3Dbox"].weight, l["3Dbox"].sigma)
        for key, name in (("cls", "cls"), ("mask", "mask")):
            if name in l:
                out[key] = (l[name].alpha, l[name].gamma)
        return out

    def _get_engine(self, B, H, W, train):
        """The launch plan for one (batch, height, width, train) key.  Plans are CACHED: the training engine -- which owns the
        Adam moments and the step count -- survives a predict_on_batch at another batch size (the epoch-end evaluation of
        callbacks/eval.py through RedirectModel), and the weights travel between plans device-to-device.  When the training
        shape itself changes, the moments and the step count move to the new plan as well."""
        key = (int(B), int(H), int(W), bool(train))
        engines = self._engines
        e = engines.get(key)
        if e is None and not train:
            e = engines.get((key[0], key[1], key[2], True))  # a training plan of the same shape can predict
        cur = self._engine
        if e is None:
            lr = self._optimizer.lr if self._optimizer else 1e-5
            clip = self._optimizer.clipnorm if self._optimizer else 0.001
            world = 1
            if train:
                from ..parallel import ensure_process_group
                world = ensure_process_group()
            e = Engine(default_context(), self.num_classes, B, H, W, self.backbone_name, self._weights,
                       train=train, lr=lr, clipnorm=clip, freeze_backbone=self.freeze_backbone, pyramid=self.pyramid,
                       anchor_params=self.anchor_params, loss_params=self._loss_params())
            if self._optimizer is not None:
                e.beta1, e.beta2, e.eps = self._optimizer.beta_1, self._optimizer.beta_2, self._optimizer.epsilon
            if train and world > 1:
                from ..parallel import DataParallel
                DataParallel(e)
            if train:  # a new training shape inherits the optimizer state of the previous training plan
                prev = next((o for k, o in reversed(list(engines.items())) if k[3]), None)
                if prev is not None:
                    e.params.m.copy_(prev.params.m)
                    e.params.v.copy_(prev.params.v)
                    e.step_count = prev.step_count
                elif self._pending_opt is not None:  # resumed from a full-model snapshot (models.load_model)
                    e.load_optimizer_state(self._pending_opt)
                    self._pending_opt = None
            e._weights_version = -1
            engines[key] = e
            # least recently used first out -- but never the plan in use, the one holding the freshest weights, or the most
            # recent TRAINING plan: it is the only holder of the Adam moments and the step count (an evaluation pass over many
            # image sizes between two training steps must not reset the optimizer)
            keep_train = next((o for k, o in reversed(list(engines.items())) if k[3]), None)
            while len(engines) > self.MAX_ENGINES:
                for k in list(engines):
                    if engines[k] is not e and engines[k] is not cur and engines[k] is not keep_train:
                        engines.pop(k).close()
                        break
                else:
                    break
        else:
            k_found = next(k for k, o in engines.items() if o is e)
            engines.move_to_end(k_found)
        if cur is not None and cur is not e and e._weights_version != self._weights_version:
            # the freshest weights live in `cur`: hand them over on the device (master + BN-folded copies + BN scales)
            torch.cuda.synchronize()
            e.params.w_master.copy_(cur.params.w_master)
            e.params.w_eff.copy_(cur.params.w_eff)
            e.params.scales.copy_(cur.params.scales)
            e.params._bn_raw = getattr(cur.params, "_bn_raw", {})
            e.refresh_planes()
        e._weights_version = self._weights_version
        self._engine = e
        return e

    def _drop_engines(self):
        if self._engine is not None:
            self._weights = self._engine.params.export()
        for e in self._engines.values():
            e.close()
        self._engines.clear()
        self._engine = None

    # ---- Keras surface -----------------------------------------------------------------------------
    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: " + name)

    def summary(self, print_fn=print):
        n = sum(v.size for k, v in self.get_weights_dict().items() if k.endswith("kernel") or k.endswith("bias"))
        print_fn("Model: %s (%s, %d classes) -- %d conv layers, %.2f M parameters, outputs %s" %
                 (self.name, self.backbone_name, self.num_classes, len(self.layers), n / 1e6, self.output_names))

    def compile(self, loss=None, optimizer=None, **kwargs):
        loss = loss or {}
        want = {"3Dbox": "orthogonal_l1", "cls": "focal", "mask": "focal"}
        for k, kind in want.items():
            l = loss.get(k)
            if l is None or getattr(l, "kind", None) != kind:
                raise ValueError("compile: output '%s' needs pyrapose_amd.losses.%s() (bin/train.py:95-102)" % (k, kind))
        self._loss = loss
        self._optimizer = optimizer or optimizers.Adam(lr=1e-5, clipnorm=0.001)
        self._drop_engines()  # a (re)compiled model starts with fresh optimizer state, like Keras
        self._pending_opt = None

    def get_weights_dict(self):
        return self._engine.params.export() if self._engine is not None else self._weights

    def load_weights(self, filepath, by_name=False, skip_mismatch=False):
        """``.npz`` written by save_weights (keys = '<layer>/kernel' HWIO, '<layer>/bias', '<bn>/...').  Keras
        ``.h5`` files need h5py, which this image does not have: convert them with
        ``python -c "import h5py, numpy ..."`` on a machine that does."""
        from ..utils.hdf5_lite import is_hdf5
        if is_hdf5(filepath):  # a real Keras / HDF5 file (whatever its name; the superblock may sit behind a user block)
            # no h5py / libhdf5 in this image: the subset reader of utils/hdf5_lite.py (version-0 superblock, old-style groups,
            # contiguous datasets -- what Keras 2.3.1 writes) + the Keras -> tensor name mapping.  Written to the HDF5
            # specification, not verified against libhdf5 output: tools/h5_to_npz.py (h5py) is the reference route.
            from ..utils import hdf5_lite, keras_names
            try:
                layers = hdf5_lite.read_keras_weights(filepath)
                data_dict = keras_names.keras_to_tensors(layers, partial=bool(by_name))
            except (hdf5_lite.H5Unsupported, ValueError, KeyError, IndexError, struct_error) as e:
                raise ImportError("load_weights: %s is an HDF5 file that the built-in subset reader cannot take (%s: %s); convert it "
                                  "where it was made with `python tools/h5_to_npz.py model.h5 model.npz`" % (filepath, type(e).__name__, e))

            class _D(object):
                files = list(data_dict)

                def __getitem__(self_, k):
                    return data_dict[k]
            data = _D()
        else:
            data = np.load(filepath)  # the zip container of save_weights -- also under the '.h5' names of ModelCheckpoint
        W = OrderedDict(self.get_weights_dict())
        for k in data.files:
            if k.startswith("optimizer/") or k.startswith("config/"):  # a full-model snapshot (save): weights only here
                continue
            if k not in W:
                if by_name:
                    continue
                raise ValueError("load_weights: unknown tensor %s" % k)
            if W[k].shape != data[k].shape:
                if skip_mismatch:
                    continue
                raise ValueError("load_weights: shape mismatch for %s: %s vs %s" % (k, W[k].shape, data[k].shape))
            W[k] = data[k].astype(np.float32)
        self._weights = W
        self._weights_version += 1
        if self._engine is not None:
            self._engine.params.load(W)
            self._engine.refresh_planes()
            self._engine._weights_version = self._weights_version

    def save_weights(self, filepath, format=None):
        """format 'h5': an HDF5 file in Keras-2.3.1's save_weights layout (utils/hdf5_lite.py writer + the name mapping of
        utils/keras_names.py) -- what the reference's own `load_weights(by_name=True)` reads.  format 'npz': the numpy
        container.  format None: by the NAME, like Keras -- '.h5' / '.hdf5' / '.keras' (the reference's per-epoch snapshots,
        bin/train.py:127-143) give HDF5, anything else the numpy container; PP_CHECKPOINT_NPZ=1 forces the container under any
        name (load_weights recognises either by content).  The file appears under its name only when complete (written to a
        temporary name beside it, then renamed)."""
        if format is None:
            ext = os.path.splitext(str(filepath))[1].lower()
            format = "h5" if (ext in (".h5", ".hdf5", ".keras") and os.environ.get("PP_CHECKPOINT_NPZ") != "1") else "npz"
        if format not in ("h5", "npz"):
            raise ValueError("save_weights: format must be 'h5', 'npz' or None, got %r" % (format,))
        tmp = "%s.tmp.%d" % (filepath, os.getpid())
        try:
            if format == "h5":
                from ..utils import hdf5_lite, keras_names
                hdf5_lite.write_keras_weights(tmp, keras_names.tensors_to_keras(self.get_weights_dict()))
            else:
                with open(tmp, "wb") as f:
                    np.savez(f, **self.get_weights_dict())
            os.replace(tmp, filepath)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    # ---- full-model snapshots (Keras `model.save`, what bin/train.py:131-142's ModelCheckpoint writes every epoch) ----------
    def _training_engine(self):
        return next((o for k, o in reversed(list(self._engines.items())) if k[3]), None)

    def _config(self):
        cfg = dict(num_classes=self.num_classes, backbone=self.backbone_name, pyramid=self.pyramid, freeze_backbone=bool(self.freeze_backbone))
        if self._optimizer is not None:
            o = self._optimizer
            cfg["optimizer"] = dict(class_name="Adam", config=dict(lr=o.lr, beta_1=o.beta_1, beta_2=o.beta_2, epsilon=o.epsilon, clipnorm=o.clipnorm))
        if self._loss is not None:
            cfg["loss"] = {k: dict(kind=l.kind, **{a: getattr(l, a) for a in ("weight", "sigma", "alpha", "gamma") if hasattr(l, a)})
                           for k, l in self._loss.items()}
        return cfg

    def optimizer_state(self):
        """Adam's state (iterations, first / second moments per trainable tensor, Keras layout) or None before the first step"""
        if self._pending_opt is not None:
            return self._pending_opt
        te = self._training_engine()
        if te is None or te.step_count == 0:
            return None
        return te.export_optimizer_state()

    def save(self, filepath, overwrite=True, include_optimizer=True, format=None):
        """Keras `model.save(filepath)`: weights AND the optimizer's state, so that `models.load_model(filepath)` resumes training
        where it stopped (bin/train.py:336-343 `--snapshot`).  HDF5 in the layout of keras/engine/saving.py (2.3.1): group
        'model_weights' (= save_weights), group 'optimizer_weights' with attribute `weight_names` = ['Adam/iterations:0',
        'training/Adam/m_<i>:0' ..., 'training/Adam/v_<i>:0' ..., 'training/Adam/vhat_<i>:0' ...] (Keras' own order: iterations,
        first moments, second moments, the (1,)-shaped amsgrad placeholders -- i over the trainable tensors in file order), root
        attributes `training_config` (optimizer class / config, loss names: Keras' JSON) and `pyrapose_amd_config` (what this
        package needs to rebuild the model: classes, backbone, pyramid, loss hyper-parameters) plus `pyrapose_amd_optimizer_tensors`
        (the tensor each m_<i> belongs to, so that loading never depends on an order).  Keras' `model_config` (the JSON of the
        layer graph) is NOT written -- it cannot be produced without Keras -- so Keras itself can `load_weights` this file but not
        `load_model` it.  Other extensions / PP_CHECKPOINT_NPZ=1: the numpy container with the same content."""
        import json
        if not overwrite and os.path.exists(filepath):
            raise IOError("save: %s exists" % filepath)
        if format is None:
            ext = os.path.splitext(str(filepath))[1].lower()
            format = "h5" if (ext in (".h5", ".hdf5", ".keras") and os.environ.get("PP_CHECKPOINT_NPZ") != "1") else "npz"
        state = self.optimizer_state() if include_optimizer else None
        cfg = self._config()
        W = self.get_weights_dict()
        tmp = "%s.tmp.%d" % (filepath, os.getpid())
        try:
            if format == "h5":
                from ..utils import hdf5_lite, keras_names
                ow, tnames = None, []
                if state is not None:
                    tnames = list(state["m"])
                    ow = OrderedDict([("Adam/iterations:0", np.array(state["iterations"], np.int64))])
                    for i, n in enumerate(tnames):
                        ow["training/Adam/m_%d:0" % i] = state["m"][n]
                    for i, n in enumerate(tnames):
                        ow["training/Adam/v_%d:0" % i] = state["v"][n]
                    for i, n in enumerate(tnames):
                        ow["training/Adam/vhat_%d:0" % i] = np.zeros((1,), np.float32)
                tc = dict(optimizer_config=cfg.get("optimizer", {}), loss={k: v["kind"] for k, v in cfg.get("loss", {}).items()}, metrics=[],
                          weighted_metrics=None, sample_weight_mode=None, loss_weights=None)
                attrs = OrderedDict(training_config=np.bytes_(json.dumps(tc).encode("utf-8")),
                                    pyrapose_amd_config=np.bytes_(json.dumps(cfg).encode("utf-8")))
                if tnames:
                    attrs["pyrapose_amd_optimizer_tensors"] = np.array([n.encode("utf-8") for n in tnames], dtype="S")
                hdf5_lite.write_keras_model(tmp, keras_names.tensors_to_keras(W), ow, attrs)
            elif format == "npz":
                extra = {"config/json": np.frombuffer(json.dumps(cfg).encode("utf-8"), np.uint8)}
                if state is not None:
                    extra["optimizer/iterations"] = np.array(state["iterations"], np.int64)
                    for n in state["m"]:
                        extra["optimizer/m/" + n] = state["m"][n]
                        extra["optimizer/v/" + n] = state["v"][n]
                with open(tmp, "wb") as f:
                    np.savez(f, **dict(W, **extra))
            else:
                raise ValueError("save: format must be 'h5', 'npz' or None, got %r" % (format,))
            os.replace(tmp, filepath)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    def set_optimizer_state(self, state):
        """Adam state as optimizer_state() returns it: applied to the training plan (now, or when the first one is built)."""
        te = self._training_engine()
        if te is not None:
            te.load_optimizer_state(state)
            self._pending_opt = None
        else:
            self._pending_opt = state

    def predict_on_batch(self, x):
        x = np.ascontiguousarray(x, np.float32)
        B, H, W, _ = x.shape
        eng = self._get_engine(B, H, W, train=False)
        eng.forward(torch.from_numpy(x).cuda())
        box, cls, mask = eng.export_outputs()
        return [box.cpu().numpy(), cls.cpu().numpy(), mask.cpu().numpy()]

    def train_on_batch(self, x, y):
        """x (B,H,W,3) float32; y = [regression_3D (B,N,17), labels (B,N,C+1), mask (B,M,C+1)] (numpy or cuda tensors).
        Returns [total, 3Dbox, cls, mask] like Keras."""
        if self._loss is None:
            raise RuntimeError("train_on_batch before compile()")
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            from ..parallel import ensure_process_group
            ensure_process_group()  # selects this rank's GPU before the first .cuda() below
        xt = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()
        B, H, W, _ = xt.shape
        eng = self._get_engine(B, H, W, train=True)
        ys = [t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t, np.float32)).cuda() for t in y]
        eng.train_step(xt, ys)
        self._weights_version += 1
        eng._weights_version = self._weights_version
        l = eng.losses()
        return [l["total"], l["3Dbox"], l["cls"], l["mask"]]

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None, workers=1,
                      use_multiprocessing=False, max_queue_size=10, **kwargs):
        """bin/train.py:381-390.  The generator follows the Sequence contract of preprocessing/generator.py:384-398."""
        callbacks = callbacks or []
        steps = int(steps_per_epoch or len(generator))
        # data parallel (WORLD_SIZE ranks, one per GPU): rank r takes batches r, r + world, ... of the generator's order, so an
        # epoch still visits every batch once and a step consumes `world` of them (the global batch)
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        if world > 1:
            from ..parallel import ensure_process_group
            ensure_process_group()  # this rank's GPU is current before the prefetcher or any .cuda() allocates
            steps = -(-steps // world)
        pick = (lambda i: generator[(i * world + rank) % len(generator)])
        history = {"loss": []}
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin()
        for epoch in range(epochs):
            run = 0.0
            if workers and torch.cuda.is_available():
                # Keras' enqueuer (workers > 0): batches are produced, pinned and uploaded ahead of the step by a background
                # thread (pyrapose_amd/prefetch.py); the losses stay on the device and are read when they are printed
                from ..prefetch import DevicePrefetcher
                acc, eng = None, None
                feed = DevicePrefetcher(pick, steps, depth=min(max(int(max_queue_size), 2), 4))
                for i, (x, y) in enumerate(feed):
                    if self._loss is None:
                        raise RuntimeError("fit_generator before compile()")
                    eng = self._get_engine(x.shape[0], x.shape[1], x.shape[2], train=True)
                    eng.train_step(x, list(y))
                    self._weights_version += 1
                    eng._weights_version = self._weights_version
                    acc = eng.loss_sums.clone() if acc is None else acc.add_(eng.loss_sums)
                    if verbose and (i % 10 == 0 or i + 1 == steps):
                        l = eng.losses()
                        print("epoch %d step %d/%d - loss: %.4f - 3Dbox: %.4f - cls: %.4f - mask: %.4f" %
                              (epoch + 1, i + 1, steps, l["total"], l["3Dbox"], l["cls"], l["mask"]))
                run = float(acc.sum().cpu()) if acc is not None else 0.0
            else:
                for i in range(steps):
                    x, y = pick(i)
                    out = self.train_on_batch(x, y)
                    run += out[0]
                    if verbose and (i % 10 == 0 or i + 1 == steps):
                        print("epoch %d step %d/%d - loss: %.4f - 3Dbox: %.4f - cls: %.4f - mask: %.4f" %
                              (epoch + 1, i + 1, steps, out[0], out[1], out[2], out[3]))
            logs = {"loss": run / max(steps, 1)}
            history["loss"].append(logs["loss"])
            if hasattr(generator, "on_epoch_end"):
                generator.on_epoch_end()
            # data parallel: the replicas are identical, so the epoch-end work (snapshots, evaluation, LR schedule on the logs it
            # produces) runs on rank 0 alone -- N ranks writing the same snapshot path at once would corrupt it; the others wait
            # at the barrier and take rank 0's learning rate and stop flag
            if world > 1:
                from ..parallel import epoch_end_sync

                def rank0_section():
                    for cb in callbacks:
                        if hasattr(cb, "on_epoch_end"):
                            cb.on_epoch_end(epoch, logs)

                def take(state):
                    if self._optimizer is not None and state[0] != self.lr:
                        self.set_lr(state[0])
                    self.stop_training = state[1]

                # (gloo side group with a long timeout: an evaluation longer than the RCCL watchdog no longer aborts the job, and
                # an exception in a rank-0 callback is re-raised on every rank instead of leaving the others in a barrier)
                epoch_end_sync(rank0_section, lambda: [float(self.lr or 0.0), bool(self.stop_training)], take)
            else:
                for cb in callbacks:
                    if hasattr(cb, "on_epoch_end"):
                        cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end()
        return history

    # learning-rate access for ReduceLROnPlateau-style callbacks
    @property
    def lr(self):
        return self._optimizer.lr if self._optimizer else None

    def set_lr(self, lr):
        self._optimizer.lr = float(lr)
        for e in self._engines.values():
            e.lr = float(lr)


class PredictionModel(object):
    """models/retinanet.py:302-335 retinanet_bbox: training model + Anchors + RegressBoxes3D;
    predict_on_batch -> [boxes3D (B,N,16), classification (B,N,C), mask (B,HW/64,C)]."""

    def __init__(self, model, anchor_params=None, name="retinanet-bbox"):
        self.model, self.name, self.anchor_params = model, name, anchor_params
        self.output_names = ["boxes3D", "cls", "mask"]

    def predict_on_batch(self, x):
        x = np.ascontiguousarray(x, np.float32)
        B, H, W, _ = x.shape
        eng = self.model._get_engine(B, H, W, train=False)
        b, c, m = eng.predict_on_batch(torch.from_numpy(x).cuda())
        return [b.cpu().numpy(), c.cpu().numpy(), m.cpu().numpy()]

    def predict_on_batch_device(self, x_dev):
        B, H, W, _ = x_dev.shape
        return self.model._get_engine(B, H, W, train=False).predict_on_batch(x_dev)


class ReduceLROnPlateau(object):
    """keras.callbacks.ReduceLROnPlateau(monitor='loss', factor=0.1, patience=2, min_delta=1e-4) as configured at
    bin/train.py:144-153 (host logic only)."""

    def __init__(self, monitor="loss", factor=0.1, patience=2, verbose=1, mode="auto", min_delta=1e-4, cooldown=0, min_lr=0):
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.best, self.wait, self.cooldown_counter, self.model = float("inf"), 0, 0, None

    def set_model(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = self.model.lr
                if old > self.min_lr:
                    new = max(old * self.factor, self.min_lr)
                    self.model.set_lr(new)
                    if self.verbose:
                        print("Epoch %05d: ReduceLROnPlateau reducing learning rate to %s." % (epoch + 1, new))
                    self.cooldown_counter, self.wait = self.cooldown, 0
