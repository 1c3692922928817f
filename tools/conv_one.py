"""Dev tool: run one conv shape repeatedly (for rocprofv3 --pmc runs).  PREC=fp32|bf16|bf16x3"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import ops
prec = os.environ.get("PREC", "fp32")
th = os.environ.get("TILE"); th = (int(th.split("x")[0]) << 16 | int(th.split("x")[1])) if th else 0
st = int(os.environ.get("STAGE", "0"))
shapes = {"ideal": (8, 128, 128, 1024, 1024, 1), "ideal3": (4, 128, 128, 256, 512, 3), "big": (4, 256, 256, 256, 256, 3), "exit": (20, 25, 25, 1536, 2048, 1), "mid": (20, 25, 25, 728, 728, 1), "rpn": (4, 46, 83, 1024, 1024, 3)}
for name in sys.argv[1:] or ["big"]:
    n, h, w, cin, cout, k = shapes[name]
    x = ops.NHWC.alloc(n, h, w, cin); x.buf.normal_()
    pk = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout), precision=prec)
    out = ops.NHWC.alloc(n, h, w, cout)
    for _ in range(4): ops.conv2d(x, pk, out, pad=(k // 2,) * 2, act=ops.ACT_RELU, tile_hint=th, stage_k=st, split_k=-1)
    torch.cuda.synchronize()
