"""Flow stage driver: the counterpart of optical_flow_net-PWC-Net/script_pwc_multi.py.

Reference boundary kept:
  * ``calculate_flow(net, im1_fn, im2_fn) -> flo[H,W,2] float32``            (script_pwc_multi.py:33-70)
  * ``writeFlowFile(filename, uv)``: tag 202021.25f, int32 W, int32 H, H*W*2 f32   (:16-31)
  * ``main``: seq_to_run.txt, weights/PReMVOS_weights/optical_flow_net/pwc_net.pth.tar,
    output/intermediate/flow/<seq>/<frame>.flo named by the first frame of the pair  (:72-103)

What changed: the per-pair host work (two cv2.resize, BGR//255, H2D, D2H, two more cv2.resize)
runs on the GPU inside the same captured graph as the network, for a batch of pairs at a time
(``FlowStage``); only uint8 frames go in and the final .flo payloads come out.
"""
from __future__ import annotations

import glob
import os
import sys
from math import ceil
from time import time
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib, jpeg, ops
from .pwcnet import PWCDCNet, pwc_dc_net

TAG_FLOAT = 202021.25


def flo_bytes(uv: np.ndarray) -> bytes:
    """The bytes of a Middlebury .flo file: tag 202021.25f, int32 W, int32 H, H*W*2 float32 (script_pwc_multi.py:16-31)."""
    uv = np.ascontiguousarray(uv, dtype=np.float32)
    if uv.ndim != 3 or uv.shape[2] != 2:
        raise ValueError("writeFlowFile: flow must have two bands!")
    return (np.array(TAG_FLOAT, dtype=np.float32).tobytes() + np.array(uv.shape[1], dtype=np.int32).tobytes()
            + np.array(uv.shape[0], dtype=np.int32).tobytes() + uv.tobytes())


def writeFlowFile(filename: str, uv: np.ndarray) -> None:
    """Middlebury .flo writer (script_pwc_multi.py:16-31)."""
    data = flo_bytes(uv)
    with open(filename, "wb") as f:
        f.write(data)


def write_flo_raw(filename: str, uv: np.ndarray) -> None:
    """``writeFlowFile``'s bytes without building them first: the 12-byte header and the array's own memory in ONE ``writev`` (no
    3.3 MB concatenation + copy per 480p file -- the merge rank of a gathered 8-GPU job writes ~430 of them per second)."""
    if uv.ndim != 3 or uv.shape[2] != 2:
        raise ValueError("writeFlowFile: flow must have two bands!")
    if uv.dtype != np.float32 or not uv.flags.c_contiguous:
        uv = np.ascontiguousarray(uv, dtype=np.float32)
    head = np.array(TAG_FLOAT, dtype=np.float32).tobytes() + np.array([uv.shape[1], uv.shape[0]], dtype=np.int32).tobytes()
    body = memoryview(uv).cast("B")
    fd = os.open(filename, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        done = os.writev(fd, [head, body])
        while done < len(head) + len(body):            # (a short write: finish the rest)
            done += os.write(fd, head[done:] if done < len(head) else body[done - len(head):])
    finally:
        os.close(fd)


def readFlowFile(filename: str) -> np.ndarray:
    """Reader with the consumer's semantics (MergeTrack/merge_functions.py:197-207)."""
    with open(filename, "rb") as f:
        tag = np.frombuffer(f.read(4), np.float32)[0]
        if tag != np.float32(TAG_FLOAT):
            raise ValueError("bad .flo tag")
        w = int(np.frombuffer(f.read(4), np.int32)[0])
        h = int(np.frombuffer(f.read(4), np.int32)[0])
        return np.frombuffer(f.read(h * w * 8), np.float32).reshape(h, w, 2).copy()


class FlowStage:
    """uint8 frame pairs [B,H,W,3] (RGB, on the GPU) -> flow [B,H,W,2] fp32 (the .flo payload)."""

    def __init__(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, batch: int = 1,
                 device=None, net: Optional[PWCDCNet] = None, use_graph: bool = True,
                 precision: Optional[str] = None):
        _lib.require_gpu()
        self.net = net if net is not None else PWCDCNet(device=device, use_graph=False, precision=precision)
        if state_dict is not None:
            self.net.load_state_dict(state_dict)
        self.batch, self.device, self.use_graph = batch, (self.net.device if net is not None and device is None else _lib.resolve_device(device)), use_graph
        self._shape = None

    def _prepare(self, h: int, w: int):
        if self._shape == (h, w):
            return
        with ops.BUILD_LOCK:
            self._prepare_locked(h, w)

    def _prepare_locked(self, h: int, w: int):
        b = self.batch
        self.h_, self.w_ = int(ceil(h / 64.0) * 64), int(ceil(w / 64.0) * 64)      # :38-45
        self.plan = self.net.plan(b, self.h_, self.w_)
        self.im1 = torch.empty((b, h, w, 3), dtype=torch.uint8, device=self.device)
        self.im2 = torch.empty_like(self.im1)
        self.out = torch.empty((b, h, w, 2), dtype=torch.float32, device=self.device)
        lib, p = _lib.load(), self.plan

        def pre():
            _lib.check(lib.premvos_flow_preprocess_u8(self.im1.data_ptr(), self.im2.data_ptr(), b, h, w,
                                                      p.img.ptr, self.h_, self.w_, _lib.current_stream()), "flow_pre")

        def post():
            f2 = p.flow2_nhwc
            _lib.check(lib.premvos_flow_postprocess_f32(f2.ptr, f2.ps, b, f2.h, f2.w, self.out.data_ptr(), h, w,
                                                        self.h_, self.w_, _lib.current_stream()), "flow_post")

        self.steps = [("flow_preprocess", pre)] + list(p.core_steps) + [("flow_postprocess", post)]
        self.graph = p.capture(self.steps) if self.use_graph else None
        self._shape = (h, w)

    def run(self, im1: torch.Tensor, im2: torch.Tensor) -> torch.Tensor:
        """Returns an internal buffer (valid until the next call) -- clone to keep."""
        assert im1.dtype == torch.uint8 and im1.shape == im2.shape and im1.shape[0] == self.batch
        self._prepare(im1.shape[1], im1.shape[2])
        self.im1.copy_(im1)
        self.im2.copy_(im2)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.plan.run(self.steps)
        return self.out

    def roofline(self, im1: torch.Tensor, im2: torch.Tensor, peak_tflops: float, reps: int = 5) -> dict:
        """Live measurement for bench.py: HIP events around every launch of the dominant kernel
        (conv_igemm_f32) on the stream it is launched on; achieved = algorithmic FLOPs / time."""
        self._prepare(im1.shape[1], im1.shape[2])
        self.im1.copy_(im1)
        self.im2.copy_(im2)
        self.plan.run(self.steps)
        conv = [(n, f) for n, f in self.steps if n.startswith("conv:")]
        evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in conv]
               for _ in range(reps)]
        for r in range(reps):
            ci = 0
            for n, f in self.steps:
                if n.startswith("conv:"):
                    evs[r][ci][0].record()
                    f()
                    evs[r][ci][1].record()
                    ci += 1
                else:
                    f()
        torch.cuda.synchronize()
        tot_ms = sum(a.elapsed_time(b) for r in evs for a, b in r) / reps
        flops = sum(self.plan.flops[n] for n, _ in conv)
        nl = len(conv)
        achieved = flops / (tot_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "conv_igemm_f32_kernel (all %d launches of one step)" % nl,
                "achieved": round(achieved, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": round(achieved / peak_tflops, 4), "traffic": None,
                "flops_per_launch": round(flops / nl, 1), "avg_launch_us": round(1e3 * tot_ms / nl, 2),
                "launches_per_step": nl}


def _imread_rgb(fn: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(fn).convert("RGB"))


def calculate_flow(net, im1_fn: str, im2_fn: str) -> np.ndarray:
    """script_pwc_multi.py:33-70 for one pair.  ``net`` is a FlowStage (batch 1) or a PWCDCNet."""
    stage = net if isinstance(net, FlowStage) else _stage_for(net)
    im = [torch.from_numpy(_imread_rgb(f)[:, :, :3].copy()).unsqueeze(0).to(stage.device) for f in (im1_fn, im2_fn)]
    return stage.run(im[0], im[1])[0].cpu().numpy()


_STAGES: Dict[int, FlowStage] = {}


def _stage_for(net: PWCDCNet) -> FlowStage:
    if id(net) not in _STAGES:
        _STAGES[id(net)] = FlowStage(net=net, batch=1, device=net.device)
    return _STAGES[id(net)]


def main(argv: Optional[List[str]] = None) -> int:
    """Same relative paths as the reference script (run from the PReMVOS root, :72,86-87)."""
    argv = sys.argv[1:] if argv is None else argv
    name = argv[0] if argv else "seq_to_run.txt"
    pwc_model_fn = argv[1] if len(argv) > 1 else "weights/PReMVOS_weights/optical_flow_net/pwc_net.pth.tar"
    out = argv[2] if len(argv) > 2 else "output/intermediate/flow"
    with open(name) as f:
        folders = [ln.rstrip() for ln in f if ln.rstrip()]
    from .. import parallel
    parallel.bind_device()                      # under torch.distributed.run: one rank per GPU, rank r takes the r-th slice of
    folders = parallel.my_videos(folders)       # the video list (the reference's curr_run_num / total_to_run scheme)
    t = time()
    net = pwc_dc_net(pwc_model_fn).cuda().eval()
    batch = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "1")))      # pairs per launch list; the host (JPEG decode, .flo
                                                                          # write) dominates, so 1 is the measured optimum
    stages: Dict[int, FlowStage] = {}
    print("Model setup, in", time() - t, "seconds")
    from .. import io_pipeline as iop
    writer = iop.Writer(enabled=iop.io_threads() > 0)
    ok = False
    try:
        for vidx, video in enumerate(folders):
            images = sorted(glob.glob(video + "*"))
            root_dir = "/".join(video.split("/")[:-2])
            outs = [im.replace(root_dir, out).replace(".png", ".flo").replace(".jpg", ".flo") for im in images]
            os.makedirs(video.replace(root_dir, out), exist_ok=True)
            t = time()
            pairs = list(zip(images[:-1], images[1:], outs))
            # frames are decoded ahead on a thread pool (every frame once), the .flo files are written by a background thread;
            # the main thread only feeds the GPU.  Same bytes as the serial loop of the reference (:94-102).
            decoded = iop.prefetch(images, jpeg.loader())       # (PREMVOS_GPU_JPEG=1: entropy decode here, the rest on the GPU)
            frames: Dict[str, object] = {}

            def frame(fn):
                while fn not in frames:
                    k = images[len(frames) + frame.dropped]
                    fr = next(decoded)
                    # a frame is the second image of one pair and the first of the next: a GPU-decoded frame is finished once
                    frames[k] = jpeg.to_device(fr) if isinstance(fr, jpeg.Decoded) else fr
                return frames[fn]
            frame.dropped = 0
            for s0 in range(0, len(pairs), batch):
                chunk = pairs[s0:s0 + batch]
                for a, b_, _ in chunk:
                    frame(a), frame(b_)
                same = all(frames[a].shape == frames[chunk[0][0]].shape and frames[b_].shape == frames[chunk[0][0]].shape
                           for a, b_, _ in chunk)
                groups = [chunk] if same else [[c] for c in chunk]
                for g in groups:
                    if len(g) not in stages:
                        stages[len(g)] = FlowStage(net=net, batch=len(g))
                    st = stages[len(g)]
                    im1 = jpeg.stack_frames([frames[a] for a, _, _ in g], st.device)
                    im2 = jpeg.stack_frames([frames[b_] for _, b_, _ in g], st.device)
                    flo = st.run(im1, im2).cpu().numpy()
                    for k, (_, _, flow_fn) in enumerate(g):
                        writer.submit(writeFlowFile, flow_fn, flo[k])
                for a, _, _ in chunk:                       # only the second image of the last pair is needed again
                    if a in frames:
                        del frames[a]
                        frame.dropped += 1
            n = max(len(images) - 1, 1)
            print("video", vidx, "finished in", time() - t, "seconds.", n, "images at", (time() - t) / n, "per image.")
        ok = True
    finally:
        try:                                    # queued .flo files are written even when a later frame failed
            writer.close()
        except BaseException:                   # noqa: BLE001 -- a writer error must not replace the error that got us here
            if ok:
                raise
    return 0


if __name__ == "__main__":
    sys.exit(main())
