"""Groundwork for the Winograd lead: error amplification of F(2x2, 3x3) relative to the direct convolution when both are
evaluated in the same finite arithmetic -- float32 as the stand-in, against float64 truth, on head-like data (ReLU activations,
N(0, 0.01)-scaled weights, 256 input channels).  CPU only (numpy)."""
import numpy as np

rng = np.random.default_rng(0)
Cin, Cout, H, W = 256, 64, 16, 16
x = np.maximum(rng.standard_normal((H + 2, W + 2, Cin)), 0)
x[0] = x[-1] = 0; x[:, 0] = x[:, -1] = 0
w = rng.standard_normal((3, 3, Cin, Cout)) * 0.02
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)


def direct(x, w, dt):
    x, w = x.astype(dt), w.astype(dt)
    y = np.zeros((H, W, Cout), dt)
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("hwc,co->hwo", x[ky:ky + H, kx:kx + W], w[ky, kx]).astype(dt)
    return y


def winograd(x, w, dt):
    x, w = x.astype(dt), w.astype(dt)
    U = np.einsum("ik,klco,jl->ijco", G.astype(dt), w, G.astype(dt)).astype(dt)          # 4x4 transformed filters
    y = np.zeros((H, W, Cout), dt)
    for ty in range(0, H, 2):
        for tx in range(0, W, 2):
            d = x[ty:ty + 4, tx:tx + 4]                                                     # 4x4 patch (padding included)
            V = np.einsum("ik,klc,jl->ijc", Bt.astype(dt), d, Bt.astype(dt)).astype(dt)
            M = np.einsum("ijc,ijco->ijo", V, U).astype(dt)                                 # 16 channel contractions
            y[ty:ty + 2, tx:tx + 2] = np.einsum("ik,klo,jl->ijo", At.astype(dt), M, At.astype(dt)).astype(dt)
    return y


ref = direct(x, w, np.float64)
assert np.abs(winograd(x, w, np.float64) - ref).max() < 1e-12 * np.abs(ref).max()
for name, fn in (("direct", direct), ("winograd F(2x2,3x3)", winograd)):
    e = fn(x, w, np.float32).astype(np.float64) - ref
    print("%-22s float32 vs float64: rel-L2 %.3e  max|err|/max|ref| %.3e" % (name, np.linalg.norm(e) / np.linalg.norm(ref), np.abs(e).max() / np.abs(ref).max()))
