"""Dev tool (round 5): RoIAlign per-bin kernel vs the row-walking kernel (PREMVOS_ROI_ALIGN=bin|row), same box, 16 images x 100 RoIs
on the 47x84x1024 feature map of a 749x1333 frame: bench-like boxes (random-weight RPN output: large) and DAVIS-like object boxes.
Prints time, algorithmic GB/s (80 MB written + the feature map read once) and a digest of the output (must agree)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker():
    import numpy as np, torch
    from premvos_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(0)
    B, R, fh, fw, C = 16, 100, 47, 84, 1024
    fm = ops.NHWC(torch.randn((B, fh, fw, C), device="cuda"), c=C)
    out = ops.NHWC.alloc(B * R, 14, 14, C)
    cnt = torch.full((B,), R, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(5)
    for name, lo, hi in (("object-sized boxes (60 ... 250 px)", 60, 250), ("large boxes (300 ... 1200 px)", 300, 1200),
                         ("mixed (40 ... 1333 px)", 40, 1333)):
        wh = np.minimum(rng.uniform(lo, hi, (B, R, 2)), [1333, 749])
        xy = rng.uniform(0, 1, (B, R, 2)) * (np.array([1333, 749]) - wh)
        rois = torch.tensor(np.concatenate([xy, xy + wh], -1), dtype=torch.float32).cuda()
        run = lambda: _lib.check(lib.premvos_roi_align_f32(fm.ptr, fm.ps, B, fh, fw, C, rois.data_ptr(), cnt.data_ptr(), R, 1.0 / 16, 14,   # noqa: E731
                                                           out.ptr, out.ps, _lib.current_stream()), "roi_align")
        for _ in range(3):
            run()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        t = sorted(ts)[len(ts) // 2]
        by = 4.0 * (B * R * 196 * C + B * fh * fw * C)
        dig = hashlib.md5(out.buf.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"  {name:38s} {1e3 * t:8.1f} us  {by / t / 1e6:7.1f} GB/s  digest {dig}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker()
    else:
        for mode in ("bin", "row"):
            print(f"PREMVOS_ROI_ALIGN={mode}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "w"], env=dict(os.environ, PREMVOS_ROI_ALIGN=mode), check=True)
