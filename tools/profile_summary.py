"""Cross-check of bench.py's live roofline against the rocprofv3 kernel stats of the same (serial) command, as quoted in
profiles/rNN_README.md:   python tools/profile_summary.py gpurun_out r01 [steps=3 warmup=1 reps=5]"""
import csv, json, sys
d, tag = sys.argv[1], sys.argv[2]
steps, warm, reps = (int(x) for x in (sys.argv[3:6] + ["3", "1", "5"][len(sys.argv) - 3:]))
for mode in ("serial", "concurrent"):
    rows = list(csv.DictReader(open(f"{d}/{tag}_bench_fp32_{mode}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    sel = lambda key: [r for r in rows if key in r["Name"]]                      # noqa: E731
    ms = lambda rs: sum(float(r["TotalDurationNs"]) for r in rs) / 1e6           # noqa: E731
    n = lambda rs: sum(int(r["Calls"]) for r in rs)                               # noqa: E731
    conv, red, sn, dw, cal = sel("conv_igemm_f32_kernel"), sel("splitk_reduce"), sel("conv_smalln"), sel("dwconv"), sel("calibrate")
    slabs = [r for r in conv if "false, true" in r["Name"] or "true, true" in r["Name"]]
    print(f"[{mode}] GPU time {tot:.1f} ms (calibrate {ms(cal):.1f}); conv_igemm {ms(conv):.1f} ms / {n(conv)} launches "
          f"({n(slabs)} k-slab), avg {1e3 * ms(conv) / n(conv):.1f} us; splitk_reduce {ms(red):.1f} ms / {n(red)}; "
          f"conv_smalln {ms(sn):.1f} ms / {n(sn)}; dw {ms(dw):.1f} ms")
    work = tot - ms(cal)
    print(f"    shares: conv (igemm + reduce + smalln) {(ms(conv) + ms(red) + ms(sn)) / work:.3f}, depthwise {ms(dw) / work:.3f}")
    if mode == "serial":
        b = json.load(open(f"{d}/{tag}_bench_under_rocprof_serial.json"))["roofline"]
        full = 1 + warm + steps                      # plan warm-up run + warm-up steps + timed steps
        pred = full * b["conv_ms_per_step"] + reps * b["conv_ms_each_layer_once"]
        meas = ms(conv) + ms(red) + ms(sn)
        print(f"    live: {b['achieved']} TFLOP/s, {b['avg_launch_us']} us per conv step ({b['launches_per_step']} per pipeline step); "
              f"predicted conv time of the trace from the live per-layer figures {pred:.1f} ms vs rocprof {meas:.1f} ms "
              f"({100 * (meas / pred - 1):+.1f} %)")
t = json.load(open(f"{d}/{tag}_conv_hbm_traffic.json"))
b = json.load(open(f"{d}/{tag}_bench_fp32.json"))
print(f"traffic {t['hbm_bytes_per_launch'] / 1e6:.0f} MB/launch ({t['launches']} launches in the PMC passes) vs algorithmic "
      f"{b['roofline']['algorithmic_bytes_per_launch'] / 1e6:.0f} MB; bench: {b['value']} frames/s, {b['roofline']['achieved']} TFLOP/s, "
      f"frac {b['roofline']['frac']}, ceiling {b['roofline']['mfma_ceiling_measured']}, cpu {b.get('cpu_baseline', {}).get('value')}")
