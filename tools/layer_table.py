"""Dev tool: per-conv-layer time / TF/s table of the full pipeline (serial eager), sorted by time lost vs 120 TF/s."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from premvos_amd import synth
from premvos_amd.pipeline import FramePipeline
B = int(os.environ.get('PREMVOS_BENCH_BATCH', '16'))
pipe = FramePipeline(synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0), batch=B, boxes_per_frame=20)
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
for (st, name, fn, fl, _, d), sm in zip(items, samples):
    name = name + (" [winograd]" if d.tile_hint == 2 else f" [winograd, slab-free {d.stage_k}]" if d.tile_hint == 3 else " [winograd F(4x4)]" if d.tile_hint == 4 else "")
    ms = sorted(sm)[2]; mult = pipe.refine_calls_per_step if st == "refine" else 1
    per = fl / mult
    rows.append((ms * mult - fl / 120e9, st, name, ms, per / ms / 1e9, mult))
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
print("3x3 stride-1 layers (Winograd candidates):")
for l, st, name, ms, tf, mult in rows:
    if "winograd" in name:
        print(f"  {ms*1e3:8.1f} us x{mult}  {tf:6.1f} TF/s  {st}:{name}")
print("worst single launches:")
for l, st, name, ms, tf, mult in sorted(rows, reverse=True)[:25]:
    print(f"  lost {l:6.2f} ms  {ms*1e3:8.1f} us x{mult}  {tf:6.1f} TF/s  {st}:{name}")
