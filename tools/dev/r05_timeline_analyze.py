import sys, numpy as np
for name in sys.argv[1:]:
    t = np.load(f"gpurun_out/r05/timeline_{name}.npy").astype(np.int64)
    t0, t1, t2, t3, w0, hw, xcc, w3 = t.T
    xcc = xcc & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"== {name}: {len(t)} workgroups on {len(np.unique(cuid))} CU ids, xcc {np.unique(xcc)}")
    wall = (w3.max() - w0.min()) / 100.0
    print(f"   kernel wall (first entry -> last end) {wall:.1f} us")
    pro, loop, epi = (t1 - t0), (t2 - t1), (t3 - t2)
    print(f"   cycles: prologue med {np.median(pro):.0f} (p10 {np.percentile(pro,10):.0f} p90 {np.percentile(pro,90):.0f}) | loop med {np.median(loop):.0f} p10 {np.percentile(loop,10):.0f} p90 {np.percentile(loop,90):.0f} | epilogue med {np.median(epi):.0f} p10 {np.percentile(epi,10):.0f} p90 {np.percentile(epi,90):.0f}")
    dur_us = (w3 - w0) / 100.0
    print(f"   workgroup duration us: med {np.median(dur_us):.1f} p10 {np.percentile(dur_us,10):.1f} p90 {np.percentile(dur_us,90):.1f}; shader clock from wg: {np.median((t3 - t0) / np.maximum(dur_us, 1e-9)) / 1e3:.3f} GHz")
    # per CU: how many workgroups are inside their main loop at each instant (wall clock, 10 ns ticks)
    base = w0.min()
    # convert loop interval to wall ticks by linear map per wg
    scale = (w3 - w0) / np.maximum(t3 - t0, 1)
    l0 = w0 + (t1 - t0) * scale; l1 = w0 + (t2 - t0) * scale
    T = int(w3.max() - base) + 1
    hist = np.zeros(8)
    resid = np.zeros(8)
    for c in np.unique(cuid):
        m = cuid == c
        inloop = np.zeros(T + 2, dtype=np.int32); res = np.zeros(T + 2, dtype=np.int32)
        for a, b in zip((l0[m] - base).astype(int), (l1[m] - base).astype(int)):
            inloop[a] += 1; inloop[b] -= 1
        for a, b in zip((w0[m] - base).astype(int), (w3[m] - base).astype(int)):
            res[a] += 1; res[b] -= 1
        inloop = np.cumsum(inloop)[:T]; res = np.cumsum(res)[:T]
        hist += np.bincount(np.clip(inloop, 0, 7), minlength=8)
        resid += np.bincount(np.clip(res, 0, 7), minlength=8)
    hist /= hist.sum(); resid /= resid.sum()
    print("   fraction of CU-time with k workgroups in their main loop:", " ".join(f"{k}:{v:.3f}" for k, v in enumerate(hist) if v > 0.0005))
    print("   fraction of CU-time with k workgroups resident:           ", " ".join(f"{k}:{v:.3f}" for k, v in enumerate(resid) if v > 0.0005))
    # gap between a workgroup's end and the next entry on the same CU slot: approximate by sorting entries and ends
    # start-time histogram over the launch (are rounds synchronised?)
    st = ((w0 - base) / 100.0)
    hh, edges = np.histogram(st, bins=40)
    print("   entries per 1/40 of the launch:", " ".join(str(v) for v in hh))
    en = ((w3 - base) / 100.0)
    hh, _ = np.histogram(en, bins=edges)
    print("   ends    per 1/40 of the launch:", " ".join(str(v) for v in hh))
    # slot refill latency: per CU, match every entry (after the first 3) with the latest end before it that has not been matched
    gaps = []
    for c in np.unique(cuid):
        m = cuid == c
        ends = np.sort(w3[m]); ents = np.sort(w0[m])[3:]
        # k-th refill entry follows the k-th end (slots are refilled in order of release)
        k = min(len(ends), len(ents))
        gaps += list((ents[:k] - ends[:k]) / 100.0)
    gaps = np.array(gaps)
    print(f"   slot refill latency (k-th end of wave 0 -> k-th later entry on that CU), us: med {np.median(gaps):.2f} p10 {np.percentile(gaps,10):.2f} p90 {np.percentile(gaps,90):.2f} mean {gaps.mean():.2f}; wg per CU: {len(t)/256:.1f}")
