"""Dev tool (round 5): tools/dev/power_data.py as SUSTAINED runs (3 s of back-to-back launches per data class -- the 17 ms bursts of the
round-3 tool are shorter than the power controller's reaction) with the socket power / shader clock sampled alongside (bench.BoxSampler).
Same instruction stream, different switching activity: what does the data cost in watts, clock and TFLOP/s?"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()
for name, n, h, w, cin, cout in (("mid 728->728 M=100k", 160, 25, 25, 728, 728), ("K=3072 -> 768 M=98k", 96, 32, 32, 3072, 768)):
    for data in ("random", "zeros", "ones", "random_relu", "small_ints", "random"):
        x = ops.NHWC.alloc(n, h, w, cin)
        wt = torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5
        if data == "random":
            x.buf.normal_()
        elif data == "random_relu":
            x.buf.normal_().clamp_(min=0)
        elif data == "ones":
            x.buf.fill_(1.0); wt.fill_(1.0)
        elif data == "small_ints":
            x.buf.copy_(torch.randint(-3, 4, x.buf.shape).float()); wt = torch.randint(-3, 4, wt.shape).float()
        else:
            wt.zero_()
        out = ops.NHWC.alloc(n, h, w, cout)
        pk = ops.pack_conv(wt, torch.zeros(cout))
        d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=16, split_k=-1)
        for _ in range(50):
            lib.premvos_conv2d_f32(C.byref(d), st)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 0.8                     # settle
        while time.perf_counter() < t_end:
            for _ in range(50): lib.premvos_conv2d_f32(C.byref(d), st)
            torch.cuda.synchronize()
        reps = 0
        with bench.BoxSampler(0, period_s=0.05) as s:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 2.5:
                for _ in range(50): lib.premvos_conv2d_f32(C.byref(d), st)
                torch.cuda.synchronize(); reps += 50
            dt = time.perf_counter() - t0
        b = s.summary()
        print(f"{name:22s} {data:12s} {dt / reps * 1e6:8.1f} us {2.0 * n * h * w * cin * cout * reps / dt / 1e12:6.1f} TF/s  "
              f"{(b.get('socket_power_w') or {}).get('mean')} W  {(b.get('sclk_mhz_mean_of_xcds') or {}).get('mean')} MHz  power-limited {b.get('power_limited_share')}", flush=True)
        del x, out
