"""Post-process the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs, --kernel-trace only) into HBM
bytes per conv step (= one dense conv layer of the pipeline: the implicit-GEMM kernel incl. its k-slab / tail-split launches and
reduce kernel, or the Winograd GEMM + output-transform pair, or the small-N direct kernel), with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE is in KB and reports half of wide coalesced reads -> x2; WRITE_SIZE in KB, uncalibrated).
The number of pipeline passes in the profiled run is the number of flow_postprocess_kernel launches (one per pass).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <conv steps per pass> [note]"""
import csv
import json
import sys

KERNELS = ("conv_igemm_f32_kernel", "wino_gemm_kernel", "wino_output_kernel", "wino_fused_kernel", "wino4_input_kernel",
           "wino4_gemm_kernel", "wino4_output_kernel", "splitk_reduce_kernel", "conv_smalln_kernel", "conv_smalln_tile_kernel", "conv_stream_f32_kernel", "conv_pwdma_f32_kernel")


def total(path, counter):
    s, n, passes, per = 0.0, 0, 0, {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            if "flow_postprocess_kernel" in row["Kernel_Name"]:
                passes += 1
            for k in KERNELS:
                if k in row["Kernel_Name"]:
                    s += float(row["Counter_Value"])
                    n += 1
                    a = per.setdefault(k, [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    break
    return s, n, passes, per


def main():
    fetch_csv, write_csv, out, steps = sys.argv[1:5]
    steps = int(steps)
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    f, nf, pf, perf = total(fetch_csv, "FETCH_SIZE")
    w, nw, pw, perw = total(write_csv, "WRITE_SIZE")
    assert nf == nw and nf > 0 and pf == pw and pf > 0, (nf, nw, pf, pw)
    fb, wb = 2.0 * f * 1024 / (pf * steps), w * 1024 / (pw * steps)
    json.dump({"kernels": list(KERNELS), "kernel_launches": nf, "pipeline_passes": pf, "conv_steps_per_pass": steps,
               "fetch_size_kb_sum": f, "write_size_kb_sum": w,
               "fetch_bytes_per_launch_corrected_x2": fb, "write_bytes_per_launch_uncalibrated": wb,
               "hbm_bytes_per_launch": fb + wb,
               "per_kernel_mb": {k: {"fetch_x2": round(2 * perf[k][0] / 1024, 1), "write": round(perw.get(k, [0, 0])[0] / 1024, 1),
                                     "launches": perf[k][1]} for k in perf},
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace on "
                         "PREMVOS_PIPELINE_SERIAL=1 bench.py --steps 1 --warmup 1 with a pre-populated PREMVOS_TUNE_CACHE "
                         "(no autotune trial launches); bytes of every kernel of a dense conv step (GEMM / k-slab / reduce / Winograd "
                         "GEMM + output transform / small-N) summed and divided by pipeline passes x conv steps per pass, i.e. per "
                         "'launch' in bench.py's sense (one conv layer); KB*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                         "reports half of wide coalesced reads); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted. " + note},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
