"""The per-frame hot path as one object: PWC-Net flow + proposal_net (general and specific weight sets,
simple_run.sh:28-42) + refinement_net on the frame's boxes, uint8 frames in HBM -> results in HBM.

Stage outputs keep the reference's interchange semantics (they become .flo / proposal JSON / refined JSON
through the stage drivers); here they stay on the device so a rank can hand them to the merge rank with one
gather (premvos_amd.parallel).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .flow.driver import FlowStage
from .proposal.driver import ProposalStage
from .proposal.model import RESNET_NUM_BLOCK, RESULTS_PER_IM
from .refinement.model import RefinementNet


class FramePipeline:
    def __init__(self, flow_sd: Dict[str, torch.Tensor], prop_general: Dict[str, object],
                 prop_specific: Dict[str, object], refine_w: Dict[str, object], batch: int = 1,
                 device=None, boxes_per_frame: int = RESULTS_PER_IM,
                 num_blocks: Sequence[int] = RESNET_NUM_BLOCK, num_middle: int = 16, concurrent: bool = True,
                 precision: Optional[str] = None, flow_precision: Optional[str] = None):
        self.batch, self.device, self.P = batch, _lib.resolve_device(device), boxes_per_frame
        device = self.device
        self.precision = precision
        self.flow = FlowStage(flow_sd, batch=batch, device=device, precision=flow_precision or precision)
        self.prop_g = ProposalStage(prop_general, batch=batch, device=device, num_blocks=num_blocks, rgb_input=True,
                                    precision=precision)
        self.prop_s = ProposalStage(prop_specific, batch=batch, device=device, num_blocks=num_blocks, rgb_input=True,
                                    precision=precision)
        self.refine = RefinementNet(refine_w, num_middle, device, precision=precision)
        # refinement: the boxes of `refine_group` frames form one batch of the network (bigger GEMMs fill the chip
        # better); `lanes` independent workspaces let several such calls be in flight on different streams
        self.refine_group = max(1, min(batch, int(os.environ.get("PREMVOS_REFINE_GROUP", "8"))))
        n_calls = self.refine_calls_per_step = -(-batch // self.refine_group)
        self.n_refine_lanes = min(n_calls, int(os.environ.get("PREMVOS_REFINE_LANES", "2"))) if concurrent else 1
        self.masks: Optional[torch.Tensor] = None
        self.conf: Optional[torch.Tensor] = None
        # the four stages of a frame are independent: each replays its HIP graph on its own stream so that the
        # partial last wave of one kernel is filled by another stage's workgroups
        self.concurrent = concurrent = concurrent and os.environ.get("PREMVOS_PIPELINE_SERIAL") != "1"
        self.streams = [torch.cuda.Stream(device=device) for _ in range(3 + self.n_refine_lanes)] if concurrent else None

    def step(self, frames_a: torch.Tensor, frames_b: torch.Tensor, boxes_y0x0y1x1: torch.Tensor):
        """frames_*: uint8 RGB [B,H,W,3] (frame t and t+1); boxes: float [B,P,4] to refine on frame t.
        Returns dict of device tensors (views of stage buffers, valid until the next step)."""
        B, H, W, _ = frames_a.shape
        if self.masks is None or self.masks.shape != (B, self.P, H, W):
            self.masks = torch.zeros((B, self.P, H, W), dtype=torch.uint8, device=self.device)
            self.conf = torch.zeros((B, self.P), dtype=torch.float32, device=self.device)
        G = self.refine_group

        def refine_all(lane=0, lanes=1):
            for c, i in enumerate(range(0, B, G)):
                if c % lanes != lane:
                    continue
                if G == 1:
                    p = self.refine.refine(frames_a[i], boxes_y0x0y1x1[i], max_boxes=self.P, lane=lane)
                    self.masks[i].copy_(p.mask)
                    self.conf[i].copy_(p.conf)
                else:
                    g = min(G, B - i)
                    p = self.refine.refine_group(frames_a[i:i + g], boxes_y0x0y1x1[i:i + g], lane=lane)
                    self.masks[i:i + g].copy_(p.mask_g)
                    self.conf[i:i + g].copy_(p.conf_g)

        if not self.concurrent:
            flo = self.flow.run(frames_a, frames_b)
            pg = self.prop_g.run(frames_a)
            ps = self.prop_s.run(frames_a)
            refine_all()
        else:
            cur = torch.cuda.current_stream()
            res = {}
            L = self.n_refine_lanes
            jobs = [("r%d" % l, (lambda l=l: refine_all(l, L))) for l in range(L)]
            jobs += [("g", lambda: self.prop_g.run(frames_a)), ("s", lambda: self.prop_s.run(frames_a)),
                     ("f", lambda: self.flow.run(frames_a, frames_b))]
            for st, (k, fn) in zip(self.streams, jobs):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    res[k] = fn()
            for st in self.streams:
                cur.wait_stream(st)
            flo, pg, ps = res["f"], res["g"], res["s"]
        return {"flow": flo, "masks": self.masks, "conf": self.conf,
                "general_boxes": pg.final_boxes, "general_probs": pg.final_probs, "general_count": pg.final_count,
                "specific_boxes": ps.final_boxes, "specific_probs": ps.final_probs, "specific_count": ps.final_count}

    def conv_steps(self):
        """(stage, name, launch fn, algorithmic FLOPs per step, algorithmic HBM bytes per step, descriptor) of every dense-conv
        step of one pipeline step (bench roofline)."""
        from . import ops
        out = []
        G = self.refine_group
        rp = self.refine.plan(self.P, *self.masks.shape[2:], False, 0, frames=G)
        for tag, steps, plan, mult in (("flow", self.flow.steps, self.flow.plan, 1), ("prop_g", self.prop_g.steps, self.prop_g.plan, 1),
                                       ("prop_s", self.prop_s.steps, self.prop_s.plan, 1),
                                       ("refine", rp.steps, rp, self.refine_calls_per_step)):
            conv = [(n, f) for n, f in steps if n.startswith("conv:")]
            assert len(conv) == len(plan.descs)
            out += [(tag, n, f, plan.flops[n] * mult, ops.algorithmic_bytes(d) * mult, d) for (n, f), d in zip(conv, plan.descs)]
        return out
