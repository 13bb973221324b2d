"""Shared measures of the GPU parity tests (test infrastructure).

* `rows_within`: the head-output bar, per ROW.  The north star asks for head outputs "within 1e-3 relative"; a max-norm
  over the whole tensor would let a small 3D-box delta be 10 % off as long as some other row is large.  Here every row
  (one anchor's 16 box values / C class scores) is judged against its OWN magnitude:
      |got - want| <= tol * max(max|want_row|, floor),   floor = floor_frac * max|want|
  (the floor only keeps rows that are zero to rounding from dividing by nothing).
* `engine_relu_masks`: the 0/1 pattern of every ReLU of one engine evaluation in the layout oracle/model_torch.py takes
  (`relu_masks`), so that the oracle differentiates the same smooth piece of the loss as the engine did.
* `grad_errors`: relative L2 error of every gradient tensor against the oracle's (frozen-BN scale folded back, L2
  regulariser added, like the optimizer does)."""
import numpy as np
import torch


def rows_within(got, want, tol=1e-3, floor_frac=1e-2):
    """-> (ok, worst): worst = max over elements of |err| / max(row magnitude, floor), to be compared with tol."""
    got = np.asarray(got, np.float64).reshape(-1, np.shape(got)[-1])
    want = np.asarray(want, np.float64).reshape(-1, np.shape(want)[-1])
    assert got.shape == want.shape, (got.shape, want.shape)
    floor = floor_frac * max(float(np.abs(want).max()), 1e-30)
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), floor)
    worst = float((np.abs(got - want) / scale).max())
    return worst <= tol, worst


def assert_rows_within(got, want, name, tol=1e-3, floor_frac=1e-2):
    ok, worst = rows_within(got, want, tol, floor_frac)
    assert ok, "%s: worst per-row relative error %.3e > %.1e" % (name, worst, tol)
    return worst


def engine_relu_masks(eng):
    """name -> bool tensor [B, C, h, w] (or a list of them, one per pyramid level, for the shared heads)"""
    masks = {}
    for name, act in eng.acts.items():
        if not act.relu:
            continue
        t = act.f32(eng.ctx)[:, : act.C].detach().cpu()
        per_level, r0 = [], 0
        for (h, w) in act.shapes:
            n = act.n_img * h * w
            per_level.append((t[r0: r0 + n].reshape(act.n_img, h, w, act.C) > 0).permute(0, 3, 1, 2).contiguous())
            r0 += n
        is_head = name.split("_")[0] in ("reg", "cls", "mask") and "conv" in name
        masks[name] = per_level if is_head else per_level[0]
    return masks


def grad_errors(eng, g_ref, Wt):
    """-> (per-tensor dict of relative L2 errors, whole-vector relative L2 error)"""
    P = eng.params
    g_eff = P.export(P.grad)
    sc = P.scales.cpu().numpy()
    errs, num, den, parts = {}, 0.0, 0.0, {}
    for key, gr in g_ref.items():
        layer, kind = key.split("/")
        s = P.specs[layer]
        g = g_eff[key].astype(np.float64)
        if kind == "kernel":
            if s.bn:
                off = P.entries[key]["scale_off"]
                g = g * sc[off: off + s.cout][None, None, None, :]
            if s.l2:
                g = g + 2 * s.l2 * np.asarray(Wt[key], np.float64)
        ref = gr.detach().numpy().astype(np.float64)
        n_, d_ = float(((g - ref) ** 2).sum()), float((ref ** 2).sum())
        parts[key] = (n_, d_)
        num += n_
        den += d_
    # a tensor whose gradient is (next to) nothing is judged against 1e-4 of the whole vector's norm, not against itself
    for key, (n_, d_) in parts.items():
        errs[key] = float(np.sqrt(n_ / max(d_, 1e-8 * den, 1e-300)))
    return errs, float(np.sqrt(num / max(den, 1e-300)))


def assert_grads_within(eng, g_ref, Wt, tol=1e-3, what=""):
    errs, total = grad_errors(eng, g_ref, Wt)
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] <= tol, "%s gradient of %s: relative L2 error %.3e > %.1e (whole vector %.3e)" % (what, worst[0], worst[1], tol, total)
    assert total <= tol, (what, total)
    return worst, total
