"""Dev tool: F(4x4,3x3) layers of the pipeline on the four GEMM block configurations; run once per library build
(PREMVOS_LIB_PATH=...) on the same box for A/B comparisons of kernel changes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
LAYERS = [("res group2 conv2 256->256", 16, 47, 84, 256, 256), ("res group1 conv2 128->128", 16, 94, 167, 128, 128),
          ("rpn conv0 1024->1024", 16, 47, 84, 1024, 1024), ("res group3 conv2 512->512 (RoIs)", 1600, 7, 7, 512, 512),
          ("pwc conv2_1 245->128 @128x224", 16, 128, 224, 245, 128), ("pwc dc_conv1 565->128 @128x224", 16, 128, 224, 565, 128)]
for name, n, h, w, cin, cout in LAYERS:
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    pk = ops.pack_conv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, torch.zeros(cout))
    res = []
    for sk in (0, 64, 16, 80):
        d = ops.conv_desc(x, pk, out, pad=(1, 1), act=ops.ACT_RELU, tile_hint=4, stage_k=sk, split_k=-1)
        ws = torch.empty((ops.workspace_bytes(d) + 3) // 4, dtype=torch.float32, device="cuda")
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        for _ in range(3):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.premvos_conv2d_f32(C.byref(d), st)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        res.append(f"{sk:2d}: {best:7.1f} us {2.0 * n * h * w * 9 * cin * cout / best / 1e6:6.1f}")
        del ws
    print(f"{name:34s}", " | ".join(res), flush=True)
