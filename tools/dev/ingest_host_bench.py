#!/usr/bin/env python
"""Host side of the merge rank's ingest without a GPU: one synthetic 480p chunk buffer (blob masks, 20 + 20 detections) replayed as
8 ranks x ROUNDS rounds into DeviceGather.decode_round + the writer threads (numpy twins pack the buffer once).  Frames/s by writer
count -- the iteration loop for the interpreter-lock / file-system side of tools/time_merge_ingest.py.

    python tools/dev/ingest_host_bench.py [rounds] [writers ...]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from premvos_amd import io_pipeline as iop, stream  # noqa: E402
from test_cpu_parallel import _np_pack_bits, _np_rle_pool, _np_unpack_bits  # noqa: E402

H, W, B, WORLD = 480, 854, 8, 8
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
writers = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
tmp = tempfile.mkdtemp(prefix="ingest_host_")
try:
    clips = []
    d0 = os.path.join(tmp, "in", "clip0")
    os.makedirs(d0)
    for t in range(B * rounds):
        Image.fromarray(np.zeros((H, W, 3), np.uint8)).save(os.path.join(d0, f"{t:05d}.png"))
    for r in range(WORLD):
        if r:
            os.symlink(d0, os.path.join(tmp, "in", f"clip{r}"))
        clips.append(os.path.join(tmp, "in", f"clip{r}") + "/")
    pipe = type("P", (), {"batch": B})()
    plans = [[(r, 0, B * rounds)] for r in range(WORLD)]
    dg = stream.DeviceGather(pipe, clips, plans, 0, WORLD, "cpu", pack_bits=_np_pack_bits, unpack_bits=_np_unpack_bits, rle_pool=_np_rle_pool)
    rng = np.random.default_rng(0)
    r = dg.staging()
    r["hw"] = (H, W)
    r["flow"][:] = torch.from_numpy(rng.standard_normal((B, H, W, 2)).astype(np.float32))
    yy, xx = np.mgrid[:H, :W]
    m = np.zeros((B, dg.P, H, W), np.uint8)
    for i in range(B):
        for j in range(40):
            for _ in range(3):
                cy, cx, rad = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(20, 150)
                m[i, j] |= ((yy - cy) ** 2 + (xx - cx) ** 2 < rad * rad).astype(np.uint8)
    r["masks"][:] = torch.from_numpy(m)
    for key in ("general", "specific"):
        xy = rng.uniform(0, 900, (B, 20, 2)).astype(np.float32)
        r[key + "_boxes"][:] = torch.from_numpy(np.concatenate([xy, xy + rng.uniform(10, 400, (B, 20, 2)).astype(np.float32)], -1))
        r[key + "_probs"][:] = torch.from_numpy(rng.uniform(0.5, 1, (B, 20)).astype(np.float32))
        r[key + "_count"][:] = 20
    r["conf"][:] = torch.from_numpy(rng.uniform(-1, 1, (B, dg.P)).astype(np.float32))
    t = time.perf_counter()
    buf = dg.x.pack(r).clone()
    print(f"packed one chunk in {time.perf_counter() - t:.1f} s; boundaries {int(dg.x.unpack(buf, masks=False)['rle_offsets'][-1])}", flush=True)
    for nw in writers:
        out = os.path.join(tmp, f"out{nw}")
        wr = iop.Writer(threads=nw)
        t = time.perf_counter()
        t_dec = 0.0
        for k in range(dg.rounds):
            t1 = time.perf_counter()
            dg.decode_round(k, [buf] * WORLD, out, wr)
            t_dec += time.perf_counter() - t1
        wr.close()
        dt = time.perf_counter() - t
        n = WORLD * B * rounds
        print(f"writers {nw}: {n / dt:7.1f} frames/s  ({1e3 * dt / n:.2f} ms/frame; exchange thread {t_dec / dt:.2f} busy; writers {wr.busy_s / dt / nw:.2f} busy each; "
              f"{1e3 * wr.busy_s / n:.2f} ms of writer time per frame)", flush=True)
        shutil.rmtree(out, ignore_errors=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
