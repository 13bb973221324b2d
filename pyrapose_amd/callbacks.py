"""The callbacks bin/train.py:107-155 (create_callbacks) hands to fit_generator, with the Keras / reference names:
ModelCheckpoint (keras.callbacks.ModelCheckpoint as configured there: one file per epoch, '{backbone}_{dataset}_{epoch:02d}.h5'),
RedirectModel (callbacks/common.py:4-47) and ReduceLROnPlateau (monitor='loss', factor 0.1, patience 2).

Checkpoint files: like Keras, ModelCheckpoint writes FULL models unless save_weights_only is set -- PyraPoseModel.save: HDF5 in
Keras-2.3.1's `model.save` layout ('model_weights' + 'optimizer_weights' with Adam's iterations / m / v + `training_config`) under
the reference's '.h5' names, written by utils/hdf5_lite.py (no h5py in this image); `models.load_model` of such a file resumes
training with the optimizer state (bin/train.py:336-343).  Other extensions, or PP_CHECKPOINT_NPZ=1, give the numpy container
with the same content; either is recognised by content, not by extension."""
from .models.model import ReduceLROnPlateau  # noqa: F401  (re-export under the callbacks namespace)


class Callback(object):
    """keras.callbacks.Callback surface used by the reference."""

    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class ModelCheckpoint(Callback):
    """keras.callbacks.ModelCheckpoint(filepath, verbose=0, save_best_only=False, monitor='val_loss', mode='auto', period=1):
    `filepath` may contain {epoch:02d} and the keys of `logs`; with save_best_only the file is written when `monitor`
    improves (mode 'auto': 'max' for names containing 'acc' or starting with 'fmeasure', else 'min')."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False, mode="auto", period=1):
        super(ModelCheckpoint, self).__init__()
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.period, self.epochs_since_last_save = save_best_only, int(period), 0
        self.save_weights_only = bool(save_weights_only)
        if mode not in ("auto", "min", "max"):
            mode = "auto"
        if mode == "auto":
            mode = "max" if ("acc" in monitor or monitor.startswith("fmeasure")) else "min"
        self.better = (lambda a, b: a > b) if mode == "max" else (lambda a, b: a < b)
        self.best = -float("inf") if mode == "max" else float("inf")

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.better(cur, self.best):
                return
            self.best = cur
        if self.verbose:
            print("\nEpoch %05d: saving model to %s" % (epoch + 1, path))
        if self.save_weights_only:
            self.model.save_weights(path)
        else:
            self.model.save(path)


class RedirectModel(Callback):
    """callbacks/common.py:4-47: wraps another callback, executed on a different model."""

    def __init__(self, callback, model):
        super(RedirectModel, self).__init__()
        self.callback, self.redirect_model = callback, model

    def on_epoch_begin(self, epoch, logs=None):
        self.callback.on_epoch_begin(epoch, logs=logs)

    def on_epoch_end(self, epoch, logs=None):
        self.callback.on_epoch_end(epoch, logs=logs)

    def on_batch_begin(self, batch, logs=None):
        self.callback.on_batch_begin(batch, logs=logs)

    def on_batch_end(self, batch, logs=None):
        self.callback.on_batch_end(batch, logs=logs)

    def on_train_begin(self, logs=None):
        self.callback.set_model(self.redirect_model)  # overwrite the model with our custom model
        self.callback.on_train_begin(logs=logs)

    def on_train_end(self, logs=None):
        self.callback.on_train_end(logs=logs)


class PoseEval(Callback):
    """callbacks/linemod.py LinemodEval (created at bin/train.py:114-124 and wrapped in RedirectModel(evaluation,
    prediction_model)): at the end of every epoch run the pose evaluation on the validation generator with the prediction
    model and put the rates into `logs` (keys 'recall' and 'detections'; the per-class arrays stay in `self.last`)."""

    def __init__(self, generator, threeD_boxes, model_points, model_diameters, K=None, threshold=0.5, min_votes=10,
                 symmetric_classes=(), tensorboard=None, verbose=1):
        super(PoseEval, self).__init__()
        self.generator, self.verbose, self.last = generator, verbose, None
        self.kw = dict(threeD_boxes=threeD_boxes, model_points=model_points, model_diameters=model_diameters, K=K, threshold=threshold,
                       min_votes=min_votes, symmetric_classes=symmetric_classes)

    def on_epoch_end(self, epoch, logs=None):
        from .utils.eval_pose import evaluate_add
        self.last = evaluate_add(self.generator, self.model.predict_on_batch, **self.kw)
        if logs is not None:
            logs["recall"], logs["detections"] = self.last["recall_all"], self.last["detections_all"]
        if self.verbose:
            print("epoch %d: ADD(-S) recall %.4f, detections %.4f" % (epoch + 1, self.last["recall_all"], self.last["detections_all"]))
