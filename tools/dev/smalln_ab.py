import torch, sys, os
sys.path.insert(0, "/root/repo")
from premvos_amd import ops
import torch.nn.functional as F
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1000 / reps
for (n, h, w, cin) in [(16, 128, 224, 565), (16, 64, 112, 597), (16, 32, 56, 629), (2, 37, 53, 117), (16, 128, 224, 32)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((2, cin, 3, 3), generator=g) * (1.0 / (9 * cin)) ** 0.5
    b = torch.randn((2,), generator=g)
    xin = ops.NHWC.alloc(n, h, w, cin); xin.buf[..., :cin] = x.permute(0, 2, 3, 1).cuda()
    pk = ops.pack_conv(wt, b)
    outs = {}
    for sk in (1, 0):
        out = ops.NHWC.alloc(n, h, w, 2)
        d = ops.conv_desc(xin, pk, out, pad=(1, 1), act=ops.ACT_LEAKY, tile_hint=1, stage_k=sk)
        ws = ops.assign_workspace([d])
        t = timeit(lambda: ops.run_desc(d))
        outs[sk] = (out.torch().cpu(), t)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.1)
    e0 = (outs[1][0].double() - ref).abs().max().item(); e1 = (outs[0][0].double() - ref).abs().max().item()
    print(f"{n}x{h}x{w}x{cin}: per-pixel {outs[1][1]:.1f} us (err {e0:.2e})   tiled {outs[0][1]:.1f} us (err {e1:.2e})")
