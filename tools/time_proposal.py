"""Dev tool: per-step timing of the full-depth proposal_net at DAVIS shape."""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import proposal_oracle as P
from premvos_amd.proposal import ProposalStage
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = P.synth_weights(0)
st = ProposalStage(w, batch=B)
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 480, 854, 3), dtype=np.uint8)).cuda()
for _ in range(2): st.run(fr)
torch.cuda.synchronize(); t = time.time(); N = 10
for _ in range(N): st.run(fr)
torch.cuda.synchronize(); dt = (time.time() - t) / N
fl = sum(st.plan.flops.values())
print(f"proposal B={B}: {dt*1e3:.2f} ms/iter {B/dt:.1f} img/s  conv {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s; rois {st.plan.roi_count.tolist()} dets {st.plan.final_count.tolist()}")
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in st.steps]
for (a, b), (n, f) in zip(ev, st.steps):
    a.record(); f(); b.record()
torch.cuda.synchronize()
rows = [(a.elapsed_time(b), n) for (a, b), (n, f) in zip(ev, st.steps)]
print("sum eager", sum(r[0] for r in rows), "ms", len(rows), "launches")
agg = {}
for ms, n in rows:
    k = n.split('/')[0] if n.startswith('conv:group') else n
    agg[k] = agg.get(k, 0) + ms
for k, v in sorted(agg.items(), key=lambda x: -x[1])[:14]:
    fl = sum(f for n, f in st.plan.flops.items() if (n.split('/')[0] if n.startswith('conv:group') else n) == k)
    print(f"  {v:8.3f} ms  {k:28s} {fl/1e9:8.1f} GF  {fl/(v*1e-3)/1e12 if v else 0:6.1f} TF/s")
