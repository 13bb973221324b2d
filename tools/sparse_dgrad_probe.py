"""Timing probe of the sparse data gradient (pp_ctx_set_row_block_skip) on the regression-head shape with two 16x16-cell
patches of non-zero gradient per image: dense vs listed-block launch, and the live block counts.  PP_SPARSE_DGRAD=0/12/22."""
import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyrapose_amd import ops
ctx = ops.Context(0)
B, shapes, cin, cout, k = 8, [(60, 80), (30, 40), (15, 20)], 512, 512, 3
rows = sum(B*h*w for h, w in shapes)
d = ops.make_conv_desc(B, shapes, shapes, cin, cout, k, 1, 1, 1, cin, cout, cout)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((rows, cin), device="cuda", generator=g)
w = torch.randn((k*k*cin, cout), device="cuda", generator=g) * 0.02
dy = torch.randn((rows, cout), device="cuda", generator=g)
dys = torch.zeros_like(dy)
h0, w0 = shapes[0]
for n in range(B):
    for (cy, cx) in ((20, 30), (45, 60)):
        r = 8
        for yy in range(cy - r, cy + r):
            a = (n*h0 + yy)*w0
            dys[a + cx - r: a + cx + r] = dy[a + cx - r: a + cx + r]
i16 = dict(dtype=torch.int16, device="cuda")
fh, fl = torch.zeros((9, cout, cin), **i16), torch.zeros((9, cout, cin), **i16)
dh, dl = torch.zeros((9, cin, cout), **i16), torch.zeros((9, cin, cout), **i16)
ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
flags, blocks = ops.row_block_list(ctx, dys, cout)
nb = (rows + 31)//32
dx = torch.empty((rows, cin), device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0)/n*1e6
print("dense", t(lambda: ops.conv_bwd_data3(ctx, d, dys, dh, dl, None, x, dx)))
print("skip ", t(lambda: ops.conv_bwd_data3(ctx, d, dys, dh, dl, None, x, dx, dy_skip=(flags, blocks))))
print("nb", nb, "live in", int(flags[:nb].sum()), "live out", int(flags[nb:].sum()), "list0", int(blocks[0]), "outlist0", int(blocks[nb+1]))
