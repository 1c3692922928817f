"""Dev tool (round 5): what does amdsmi report while a GEMM loop runs?  (clock / power / temperature sampling for bench.py)"""
import os, sys, time, threading, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
h = hs[0]
m = amdsmi.amdsmi_get_gpu_metrics_info(h)
print({k: v for k, v in m.items() if not isinstance(v, (list, dict)) or k in ("current_gfxclks",)})
lib, st = _lib.load(), _lib.current_stream()
stop = False
samples = []
def sampler():
    while not stop:
        t = time.perf_counter()
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            samples.append((t, m.get("current_gfxclks"), m.get("current_socket_power"), m.get("temperature_hotspot"), m.get("average_gfx_activity"),
                            m.get("current_uclk"), m.get("throttle_status"), m.get("accumulation_counter"), m.get("prochot_residency_acc"),
                            m.get("ppt_residency_acc"), m.get("socket_thm_residency_acc"), m.get("vr_thm_residency_acc"), m.get("hbm_thm_residency_acc")))
        except Exception as e:
            samples.append((t, "err", str(e)))
        time.sleep(0.02)
for name, n, hh, w, cin, cout in (("mid728", 160, 25, 25, 728, 728), ("k3072", 96, 32, 32, 3072, 768)):
    x = ops.NHWC(torch.randn((n, hh, w, cin), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.zeros(cout))
    out = ops.NHWC.alloc(n, hh, w, cout)
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=16, split_k=-1)
    for _ in range(3):
        _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(50):
            lib.premvos_conv2d_f32(C.byref(d), st)
        torch.cuda.synchronize(); reps += 50
    dt = time.perf_counter() - t0
    stop = True; th.join()
    print(f"{name}: {dt / reps * 1e6:.1f} us per launch, {2.0 * n * hh * w * cin * cout * reps / dt / 1e12:.1f} TF/s, {len(samples)} samples")
    for s in samples[:: max(1, len(samples) // 12)]:
        print("   ", s)
