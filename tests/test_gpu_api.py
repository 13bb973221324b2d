"""GPU: the reference-shaped Python surface (models.backbone(...).retinanet, compile, fit_generator,
predict_on_batch, convert_model, save/load, loss functors) drives the HIP engine end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class TinyGenerator(object):
    """Sequence contract of preprocessing/generator.py:384-398 on synthetic data."""

    def __init__(self, B, H, W, C, n_batches=2, seed=0):
        from oracle import anchors_np as OA
        rng = np.random.default_rng(seed)
        self.batches = []
        anchors = OA.anchors_for_shape((H, W))
        for _ in range(n_batches):
            x = rng.integers(0, 256, size=(B, H, W, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
            anns = []
            for b in range(B):
                mask = np.zeros((H, W), np.uint8)
                bw, bh = rng.uniform(24, 60, 2)
                x1, y1 = rng.uniform(0, W - bw), rng.uniform(0, H - bh)
                mask[int(y1):int(y1 + bh), int(x1):int(x1 + bw)] = 1
                box = rng.uniform(-40, 40, size=(1, 8, 3)).astype(np.float32)
                anns.append({"mask": [mask], "labels": np.array([float(rng.integers(0, C))]),
                             "bboxes": np.array([[x1, y1, x1 + bw, y1 + bh]]),
                             "poses": np.array([[0.0, 0.0, 800.0, 1.0, 0.0, 0.0, 0.0]]), "segmentations": box.astype(np.float64),
                             "cam_params": np.array([[572.4114, 573.57043, 325.2611, 242.04899]]), "mask_ids": np.array([1.0])})
            reg, lab, msk = OA.anchor_targets_bbox(anchors, [(H, W)] * B, anns, C)
            self.batches.append((x, [reg, lab, msk]))

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        return self.batches[i]

    def on_epoch_end(self):
        pass


def test_train_and_predict_through_reference_api(tmp_path):
    from pyrapose_amd import losses, models, optimizers
    from pyrapose_amd.models.model import ReduceLROnPlateau
    B, H, W, C = 2, 96, 128, 5
    backbone = models.backbone("resnet50")
    model = backbone.retinanet(C)
    models.check_training_model(model)
    prediction_model = models.convert_model(model)
    model.compile(loss={"3Dbox": losses.orthogonal_l1(), "cls": losses.focal(), "mask": losses.focal()},
                  optimizer=optimizers.Adam(lr=1e-4, clipnorm=0.001))
    gen = TinyGenerator(B, H, W, C)
    w_before = model.get_weights_dict()["reg_out/kernel"].copy()
    hist = model.fit_generator(gen, steps_per_epoch=3, epochs=2, verbose=0, callbacks=[ReduceLROnPlateau(verbose=0)])
    assert len(hist["loss"]) == 2 and all(np.isfinite(hist["loss"]))
    w_after = model.get_weights_dict()["reg_out/kernel"]
    assert np.abs(w_after - w_before).max() > 0            # the optimizer moved the trainable weights
    frozen = model.get_weights_dict()["conv1/kernel"]
    x = gen[0][0]
    boxes3d, scores, mask = prediction_model.predict_on_batch(x)
    N = sum(h * w for h, w in [(12, 16), (6, 8), (3, 4)]) * 9
    assert boxes3d.shape == (B, N, 16) and scores.shape == (B, N, C) and mask.shape == (B, 12 * 16, C)
    assert np.isfinite(boxes3d).all() and (scores > 0).all() and (scores < 1).all()
    # save / load round trip reproduces the predictions bit for bit
    path = os.path.join(str(tmp_path), "resnet50_linemod_01.npz")
    model.save(path)
    m2 = models.load_model(path, backbone_name="resnet50")
    assert m2.num_classes == C
    assert np.array_equal(m2.get_weights_dict()["conv1/kernel"], frozen)
    b2, s2, k2 = models.convert_model(m2).predict_on_batch(x)
    assert np.array_equal(b2, boxes3d) and np.array_equal(s2, scores) and np.array_equal(k2, mask)
    with pytest.raises(ValueError):
        backbone.retinanet(C).compile(loss={"3Dbox": losses.focal(), "cls": losses.focal(), "mask": losses.focal()})
    with pytest.raises(ValueError):
        models.backbone("resnet18")


def test_loss_functors_vs_oracle():
    from oracle import model_torch as MT
    from pyrapose_amd import losses
    rng = np.random.default_rng(1)
    B, N, C = 2, 500, 7
    logits = rng.standard_normal((B, N, C)).astype(np.float32) * 3
    y = np.zeros((B, N, C + 1), np.float32)
    st = rng.choice([-1.0, 0.0, 1.0], size=(B, N), p=[0.1, 0.8, 0.1]).astype(np.float32)
    y[:, :, C] = st
    bi, ni = np.nonzero(st == 1)
    y[bi, ni, rng.integers(0, C, size=len(bi))] = 1
    got = float(losses.focal()(torch.from_numpy(y).cuda(), torch.from_numpy(logits).cuda()).cpu())
    want = float(MT.focal(torch.from_numpy(y).double(), torch.sigmoid(torch.from_numpy(logits).double())))
    assert abs(got - want) < 1e-4 * abs(want)
    yb = rng.standard_normal((B, N, 17)).astype(np.float32)
    yb[:, :, 16] = st
    pred = rng.standard_normal((B, N, 16)).astype(np.float32)
    got = float(losses.orthogonal_l1()(torch.from_numpy(yb).cuda(), torch.from_numpy(pred).cuda()).cpu())
    want = float(MT.orthogonal_l1(torch.from_numpy(yb).double(), torch.from_numpy(pred).double()))
    assert abs(got - want) < 1e-4 * abs(want)


def test_layers_and_backend_mirrors():
    from oracle import anchors_np as OA
    from pyrapose_amd import layers
    feats = torch.zeros((2, 15, 20, 256), device="cuda")
    a = layers.Anchors(size=128, stride=32)(feats)
    want = OA.anchors_for_shape_f32((480, 640))[-15 * 20 * 9:]
    assert a.shape == (2, 2700, 4) and np.array_equal(a[1].cpu().numpy(), want)
    reg = torch.randn((2, 2700, 16), device="cuda")
    got = layers.RegressBoxes3D()([a, reg]).cpu().numpy()
    assert np.array_equal(got, OA.box3d_transform_inv_f32(want[None], reg.cpu().numpy()))
    src = torch.randn((1, 17, 23, 8), device="cuda")
    up = layers.UpsampleLike()([src, torch.zeros((1, 34, 45, 8), device="cuda")])
    assert up.shape == (1, 34, 45, 8)


def test_preprocess_u8_matches_reference_preprocessing():
    """pp_preprocess_caffe_u8 == preprocess_image(mode='caffe') + compute_inputs zero padding (utils/image.py:35-62,
    preprocessing/generator.py:319-336), bit for bit, and Engine.forward_u8 == Engine.forward on the same batch."""
    import numpy as np
    import torch
    from pyrapose_amd import ops
    from pyrapose_amd.engine import Engine
    from pyrapose_amd.runtime import default_context
    ctx = default_context()
    rng = np.random.default_rng(8)
    B, H, W = 3, 64, 96
    sizes = [(64, 96), (50, 96), (64, 71)]
    u8 = np.zeros((B, H, W, 3), np.uint8)
    want = np.zeros((B, H, W, 3), np.float32)
    for b, (h, w) in enumerate(sizes):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        u8[b, :h, :w] = img
        x = img.astype(np.float32)           # the reference's three statements
        x[..., 0] -= 103.939
        x[..., 1] -= 116.779
        x[..., 2] -= 123.68
        want[b, :h, :w] = x
    x4 = torch.full((B * H * W, 4), float("nan"), dtype=torch.float32, device="cuda")
    ops.preprocess_caffe_u8(ctx, torch.from_numpy(u8).cuda(), sizes, x4)
    got = x4.cpu().numpy().reshape(B, H, W, 4)
    assert np.array_equal(got[..., :3], want) and not got[..., 3].any()
    eng = Engine(ctx, 5, B, H, W, train=False)
    eng.forward(torch.from_numpy(want).cuda())
    a = [t.clone() for t in eng.export_outputs()]
    eng.forward_u8(torch.from_numpy(u8).cuda(), sizes)
    b_ = eng.export_outputs()
    for p, q in zip(a, b_):
        assert torch.equal(p, q)
    with pytest.raises(ValueError):
        ops.preprocess_caffe_u8(ctx, torch.from_numpy(u8).cuda(), [(65, 96)] * B, x4)


def test_fit_generator_prefetch_matches_synchronous_loop():
    """fit_generator(workers=1): batches are uploaded ahead of the step by pyrapose_amd/prefetch.py and the losses stay on
    the device; the epoch losses and the trained weights must equal those of the synchronous path (workers=0)."""
    import numpy as np
    import torch
    from pyrapose_amd import losses, models, optimizers
    from pyrapose_amd.prefetch import DevicePrefetcher

    class Gen(object):  # preprocessing/generator.py:384-398 contract
        def __init__(self, n, B, H, W, C, N, M):
            rng = np.random.default_rng(3)
            self.b = []
            for _ in range(n):
                x = (rng.integers(0, 256, (B, H, W, 3)).astype(np.float32) - 110.0)
                yb = rng.standard_normal((B, N, 17)).astype(np.float32); yb[:, :, 16] = (rng.uniform(size=(B, N)) < 0.03)
                yc = np.zeros((B, N, C + 1), np.float32); st = (rng.uniform(size=(B, N)) < 0.03); yc[:, :, C] = st
                yc[:, :, 1] = st
                ym = np.zeros((B, M, C + 1), np.float32); sm = (rng.uniform(size=(B, M)) < 0.05); ym[:, :, C] = sm; ym[:, :, 0] = sm
                self.b.append((x, [yb, yc, ym]))

        def __len__(self):
            return len(self.b)

        def __getitem__(self, i):
            return self.b[i]

    B, H, W, C = 2, 64, 96, 4
    N = sum(((H + 2 ** l - 1) // 2 ** l) * ((W + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5)) * 9
    gen = Gen(5, B, H, W, C, N, ((H + 7) // 8) * ((W + 7) // 8))
    # the uploader hands out the generator's batches in order, bit for bit
    for i, (xd, yd) in enumerate(DevicePrefetcher(lambda i: gen[i % len(gen)], 7, depth=3)):
        x, ys = gen[i % len(gen)]
        assert torch.equal(xd.cpu(), torch.from_numpy(x)) and all(torch.equal(a.cpu(), torch.from_numpy(b)) for a, b in zip(yd, ys))
    assert i == 6
    results = []
    for workers in (0, 1):
        m = models.backbone("resnet50").retinanet(num_classes=C)
        m.compile(loss={"3Dbox": losses.orthogonal_l1(), "cls": losses.focal(), "mask": losses.focal()},
                  optimizer=optimizers.Adam(lr=1e-4, clipnorm=0.001))
        h = m.fit_generator(gen, steps_per_epoch=5, epochs=2, verbose=0, workers=workers, max_queue_size=3)
        results.append((h["loss"], m.get_weights_dict()))
    (l0, w0), (l1, w1) = results
    assert np.allclose(l0, l1, rtol=1e-5, atol=1e-6), (l0, l1)
    for k in w0:
        assert np.abs(w0[k] - w1[k]).max() <= 1e-6 * max(np.abs(w0[k]).max(), 1e-3), k


def test_train_step_from_annotations_equals_the_numpy_feed():
    """Engine.train_step_from_annotations (uint8 images + raw annotations, preprocessing and target assignment on the device)
    takes the same step as the reference-shaped feed: preprocess_image + compute_inputs on the host, anchor_targets_bbox."""
    import numpy as np
    import torch
    import bench
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    from pyrapose_amd.runtime import default_context
    from pyrapose_amd.utils import anchors as UA
    B, H, W, C = 2, 128, 160, 5
    rng = np.random.default_rng(41)
    _, images, anns = bench.synth_batch(B, H, W, C, seed=42, side=(24, 64))
    u8 = rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)
    Wt = arch.init_weights(C, seed=43)
    ctx = default_context()
    # reference-shaped feed: float32 BGR minus caffe means (utils/image.py:58-60), numpy targets
    x = u8.astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    anchors = UA.anchors_for_shape((H, W))
    tg = UA.anchor_targets_bbox(anchors, [np.zeros((H, W, 3), np.uint8)] * B, anns, C)
    e0 = Engine(ctx, C, B, H, W, weights=Wt, train=True)
    e0.train_step(torch.from_numpy(x).cuda(), [torch.from_numpy(t).cuda() for t in tg])
    e1 = Engine(ctx, C, B, H, W, weights=Wt, train=True)
    e1.train_step_from_annotations(torch.from_numpy(u8).pin_memory(), anns)
    l0, l1 = e0.losses(), e1.losses()
    for k in l0:  # (the loss sums are float32 atomics over ~10^5 terms: two runs of the SAME step differ by up to ~1.3e-6)
        assert abs(l0[k] - l1[k]) <= 4e-6 * max(abs(l0[k]), 1e-6), (k, l0[k], l1[k])
    w0, w1 = e0.params.w_master, e1.params.w_master
    assert float((w0 - w1).abs().max()) <= 1e-7 * float(w0.abs().max()) + 1e-9
