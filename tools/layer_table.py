"""Dev tool: per-conv-layer time / TF/s table of the full pipeline (serial eager), sorted by time lost vs 120 TF/s."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from premvos_amd import synth
from premvos_amd.pipeline import FramePipeline
B = int(os.environ.get('PREMVOS_BENCH_BATCH', '16'))
pipe = FramePipeline(synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0), batch=B, boxes_per_frame=20,
                     precision=os.environ.get('LT_PRECISION', 'fp32'), flow_precision='fp32')
fa, fb = bench.synth_frames(B, 0); fa, fb = fa.cuda(), fb.cuda(); boxes = bench.synth_boxes(B, 0).cuda()
for _ in range(2): pipe.step(fa, fb, boxes)
torch.cuda.synchronize()
items = pipe.conv_steps()
samples = [[] for _ in items]
for _ in range(5):
    evs = []
    for _, _, fn, _, _, _ in items:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(evs): samples[i].append(a.elapsed_time(b))
rows = []
membound = []      # vs 5 TB/s of algorithmic bytes (input + weights + output (+ residual)); by = bytes per step of the layer
for (st, name, fn, fl, by, d), sm in zip(items, samples):
    name = name + (" [winograd]" if d.tile_hint == 2 else f" [winograd, slab-free {d.stage_k}]" if d.tile_hint == 3 else " [winograd F(4x4)]" if d.tile_hint == 4 else "")
    ms = sorted(sm)[2]; mult = pipe.refine_calls_per_step if st == "refine" else 1
    per = fl / mult
    rows.append((ms * mult - fl / 120e9, st, name, ms, per / ms / 1e9, mult))
    membound.append((ms * mult - by / 5.0e9, st, name, ms, by / mult / ms / 1e6, per / ms / 1e9, mult, per / (by / mult), d))
tot = sum(r[3] * r[5] for r in rows); lost = sum(r[0] for r in rows)
print(f"total conv {tot:.1f} ms/step; lost vs 120 TF/s {lost:.1f} ms")
# aggregate by (stage, class)
import collections
agg = collections.defaultdict(lambda: [0.0, 0.0])
for l, st, name, ms, tf, mult in rows:
    key = st + ":" + (name.split("/")[0] if st != "refine" else "/".join(name.split("/")[:2]) if "flow" in name else name.split("/")[0])
    if st == "refine" and "xception_module" in name: key = st + ":" + name.split("/")[0].replace("conv:", "") + "/" + name.split("/")[1] + ":" + name.split("/")[-1][:16]
    agg[key][0] += l; agg[key][1] += ms * mult
for k, (l, t) in sorted(agg.items(), key=lambda x: -x[1][0])[:30]:
    print(f"  lost {l:7.2f} ms  time {t:7.2f} ms  {k}")
if os.environ.get("LT_STAGE"):          # every conv of one stage, by time (LT_STAGE=flow|prop_g|prop_s|refine)
    print(f"all conv layers of stage {os.environ['LT_STAGE']}, by time:")
    sel = [r for r in rows if r[1] == os.environ["LT_STAGE"]]
    print(f"  {sum(r[3] * r[5] for r in sel):.2f} ms/step in {len(sel)} layers")
    for l, st, name, ms, tf, mult in sorted(sel, key=lambda r: -r[3] * r[5]):
        print(f"  {ms*1e3:8.1f} us x{mult}  {tf:6.1f} TF/s  lost {l:5.2f} ms  {name}")
print("3x3 stride-1 layers (Winograd candidates):")
for l, st, name, ms, tf, mult in rows:
    if "winograd" in name:
        print(f"  {ms*1e3:8.1f} us x{mult}  {tf:6.1f} TF/s  {st}:{name}")
print("worst single launches:")
for l, st, name, ms, tf, mult in sorted(rows, reverse=True)[:25]:
    print(f"  lost {l:6.2f} ms  {ms*1e3:8.1f} us x{mult}  {tf:6.1f} TF/s  {st}:{name}")
print("layers below the fp32 ridge (FLOP per algorithmic byte < 30), by time lost vs 5 TB/s:")
mb = [r for r in membound if r[7] < 30]
print(f"  {sum(r[3] * r[6] for r in mb):.1f} ms/step in {len(mb)} layers; lost vs 5 TB/s {sum(max(r[0], 0) for r in mb):.1f} ms")
for l, st, name, ms, gbs, tf, mult, inten, d in sorted(mb, key=lambda r: -r[0])[:40]:
    print(f"  lost {l:6.2f} ms  {ms*1e3:8.1f} us x{mult}  {gbs:7.0f} GB/s  {tf:6.1f} TF/s  {inten:5.1f} F/B  tile {d.tile_hint >> 16}x{d.tile_hint & 0xffff} "
          f"M={d.n * d.ho * d.wo} {d.cin}->{d.cout} k{d.kh} s{d.sh} res={bool(d.res)}  {st}:{name}")
if os.environ.get("LT_PRECISION"):
    print("proposal net (general), time per (group, layer kind):")
    kinds = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for l, st, name, ms, tf, mult in rows:
        if st == "prop_g" and "/block" in name:
            parts = name.replace("conv:", "").split("/")
            k = (parts[0], parts[2].split(" ")[0])
            kinds[k][0] += ms; kinds[k][1] += tf * ms; kinds[k][2] += 1
    for k, (t, w, n) in sorted(kinds.items()):
        print(f"  {k[0]:7s} {k[1]:13s} x{n:2d}  {t:7.3f} ms  {w / t:6.1f} TF/s-eq")
    print("  split layers:", pipe.prop_g.plan.split_layers if hasattr(pipe.prop_g, "plan") else "?")
if os.environ.get("LT_PRECISION"):
    print("refinement net, layers by time:")
    rr = sorted([r for r in rows if r[1] == "refine"], key=lambda r: -r[3] * r[5])
    print(f"  total {sum(r[3] * r[5] for r in rr):.2f} ms/step")
    for l, st, name, ms, tf, mult in rr[:28]:
        print(f"  {ms * mult:7.3f} ms  {tf:6.1f} TF/s-eq  {name}")
