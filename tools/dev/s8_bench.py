"""Dev tool: csrc/conv_bf16x3_s8.hip per layer shape and tile, against the fp32 implicit GEMM / Winograd kernel the table picks.  TFLOP/s-equivalent = algorithmic 2 M K N / time; issued fraction of the bf16 pipe = 3x that / 2500."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
TILES = [int(t) for t in os.environ.get("S8_TILES", "0,1,2,3,4,5,6,7,8,9,10,11").split(",")]


def timeit(fn, reps=10):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) * 1000 / reps)
    return best


SHAPES = [  # name, n, h, w, cin, cout, k, pad
    ("xception mid 728->728 M=100k", 160, 25, 25, 728, 728, 1, 0),
    ("xception exit 1536->2048 M=100k", 160, 25, 25, 1536, 2048, 1, 0),
    ("xception entry 256->728 49x49", 160, 49, 49, 256, 728, 1, 0),
    ("xception entry 128->256 97x97", 160, 97, 97, 128, 256, 1, 0),
    ("resnet g2 conv1 1024->256 B16", 16, 47, 84, 1024, 256, 1, 0),
    ("resnet g2 conv3 256->1024 B16", 16, 47, 84, 256, 1024, 1, 0),
    ("resnet g2 conv2 3x3 256->256 B16", 16, 47, 84, 256, 256, 3, 1),
    ("resnet g1 conv2 3x3 128->128 B16", 16, 94, 167, 128, 128, 3, 1),
    ("rpn 3x3 1024->1024 B16", 16, 47, 84, 1024, 1024, 3, 1),
    ("conv5 3x3 512->512 1600 rois", 1600, 7, 7, 512, 512, 3, 1),
    ("conv5 conv1 2048->512 1600 rois", 1600, 7, 7, 2048, 512, 1, 0),
    ("pwc L2 conv2_1 3x3 248->128 B16", 16, 128, 224, 248, 128, 3, 1),
    # K sweep at the middle flow's M and N: time(K) = fixed (prologue + epilogue + tile rounds) + K / rate
    ("ksweep 256->728 M=100k", 160, 25, 25, 256, 728, 1, 0),
    ("ksweep 1456->728 M=100k", 160, 25, 25, 1456, 728, 1, 0),
    ("ksweep 2912->728 M=100k", 160, 25, 25, 2912, 728, 1, 0),
    ("ksweep 728->768 M=100k", 160, 25, 25, 728, 768, 1, 0),
    ("ksweep 728->768 M=131k (512 row tiles)", 128, 32, 32, 728, 768, 1, 0),
]
only = os.environ.get("S8_ONLY")
rows = []
for name, n, h, w, cin, cout, k, pad in SHAPES:
    if only and only not in name:
        continue
    g = torch.Generator().manual_seed(1)
    x = ops.NHWC.alloc(n, h, w, cin, ps=(cin + 7) // 8 * 8); x.buf.normal_()
    xs = ops.NHWC.alloc_s8(n, h, w, cin)
    ops.split8(x, xs)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    out = ops.NHWC.alloc(n, h, w, cout, ps=(cout + 7) // 8 * 8)
    out8 = ops.NHWC.alloc_s8(n, h, w, cout)
    flops = 2.0 * n * h * w * cin * cout * k * k
    pk8 = ops.pack_conv_s8(wt, torch.zeros(cout))
    row = {"layer": name, "gflop": round(flops / 1e9, 1)}
    # fp32: what the shipped table (or the rule) picks for this signature
    pk32 = ops.pack_conv(wt, torch.zeros(cout))
    d32 = ops.conv_desc(x, pk32, out, pad=(pad, pad), act=ops.ACT_RELU)
    ops.autotune([d32])
    ws = ops.assign_workspace([d32])
    t32 = timeit(lambda: lib.premvos_conv2d_f32(C.byref(d32), st))
    row["fp32_us"], row["fp32_tf"], row["fp32_hint"] = round(t32, 1), round(flops / t32 / 1e6, 1), [d32.tile_hint, d32.stage_k, d32.split_k]
    ref = out.buf.clone()
    for tile in TILES:
        for mode, o32, o8 in (("f32out", out, None), ("s8out", None, out8)):
            d = ops.conv_s8_desc(xs, pk8, o32, o8, pad=(pad, pad), act=ops.ACT_RELU)
            try:
                ops.run_s8(d, xs, pk8, o8, tile)
            except _lib.PremvosError as e:
                row[f"t{tile}_{mode}"] = str(e)[:60]
                continue
            t = timeit(lambda: ops.run_s8(d, xs, pk8, o8, tile))
            row[f"t{tile}_{mode}_us"], row[f"t{tile}_{mode}_tf"] = round(t, 1), round(flops / t / 1e6, 1)
            if mode == "f32out":
                torch.cuda.synchronize()
                row[f"t{tile}_err"] = float((out.buf - ref).abs().max() / ref.abs().max())
    best = max(v for kk, v in row.items() if kk.endswith("_tf") and kk.startswith("t"))
    row["best_s8_tf"], row["best_issued_frac_of_bf16_pipe"], row["speedup_vs_fp32"] = best, round(3 * best / 2500, 3), round(best / row["fp32_tf"], 2)
    rows.append(row)
    print(json.dumps(row), flush=True)
    del x, xs, out, out8
    torch.cuda.empty_cache()
