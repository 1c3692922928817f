"""Dev tool: run one large conv back to back for a few seconds per configuration while sampling rocm-smi (power, sclk) -- is the
fp32 MFMA path power-limited?  Prints TF/s, average power and clock per (tile, stage) configuration."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from premvos_amd import ops

def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            out.append(t)
        except Exception as e:
            out.append(str(e))
        time.sleep(0.15)

def parse(samples):
    pw, ck = [], []
    for t in samples:
        m = re.search(r"Power \(W\):\s*([\d.]+)", t)
        if m: pw.append(float(m.group(1)))
        m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", t)
        if m: ck.append(float(m.group(1)))
    return (sum(pw) / max(len(pw), 1), sum(ck) / max(len(ck), 1), len(pw))

from premvos_amd import _lib
lib = _lib.load()
sink = torch.zeros(4, device="cuda")
for blocks in (256, 512, 1024, 64):
    iters = 40000
    lib.premvos_mfma_f32_calibrate(1000, blocks, sink.data_ptr(), _lib.current_stream()); torch.cuda.synchronize()
    stop, smp = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, smp)); th.start()
    t0 = time.time(); it = 0
    while time.time() - t0 < 3.0:
        lib.premvos_mfma_f32_calibrate(iters, blocks, sink.data_ptr(), _lib.current_stream()); torch.cuda.synchronize(); it += 1
    dt = time.time() - t0
    stop.set(); th.join()
    p, c, ns = parse(smp)
    fl = blocks * 4 * iters * 16 * 4096.0 * it
    print(f"pure MFMA, {blocks} blocks: {fl / dt / 1e12:6.1f} TF/s ({fl / dt / 1e12 / min(blocks, 256) * 256 / 157.3:.3f} of peak per busy CU)  power {p:6.1f} W  sclk {c:6.0f}")
n, h, w, cin, cout, k = 8, 128, 128, 1024, 1024, 1
x = ops.NHWC.alloc(n, h, w, cin); x.buf.normal_()
pk = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout))
out = ops.NHWC.alloc(n, h, w, cout)
flops = 2.0 * n * h * w * cin * cout * k * k
print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout[-1500:])
for tile, st in (((128, 128), 32), ((128, 128), 16), ((64, 64), 16)):
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(tile[0] << 16) | tile[1], stage_k=st, split_k=-1)
    for _ in range(3): ops.run_desc(d)
    torch.cuda.synchronize()
    stop, smp = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, smp)); th.start()
    t0 = time.time(); it = 0
    while time.time() - t0 < 4.0:
        for _ in range(20): ops.run_desc(d)
        torch.cuda.synchronize(); it += 20
    dt = time.time() - t0
    stop.set(); th.join()
    p, c, ns = parse(smp)
    print(f"tile {tile} stage {st}: {flops * it / dt / 1e12:6.1f} TF/s  avg power {p:6.1f} W  sclk {c:6.0f} MHz  ({ns} samples)")
    if smp: print("   last sample:", " | ".join(l.strip() for l in smp[-1].splitlines() if "Power" in l or "sclk" in l or "Temp" in l)[:400])
