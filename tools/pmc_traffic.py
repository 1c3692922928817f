"""Post-process the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs, --kernel-trace only) into HBM
bytes per launch of the conv kernel, with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE is in KB and reports
half of wide coalesced reads -> x2; WRITE_SIZE in KB, uncalibrated).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [note]"""
import csv
import json
import sys

KERNEL = "conv_igemm_f32_kernel"


def total(path, counter):
    s, n = 0.0, 0
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and KERNEL in row["Kernel_Name"]:
                s += float(row["Counter_Value"])
                n += 1
    return s, n


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    f, nf = total(fetch_csv, "FETCH_SIZE")
    w, nw = total(write_csv, "WRITE_SIZE")
    assert nf == nw and nf > 0, (nf, nw)
    fb, wb = 2.0 * f * 1024 / nf, w * 1024 / nw
    json.dump({"kernel": KERNEL, "launches": nf, "fetch_size_kb_sum": f, "write_size_kb_sum": w,
               "fetch_bytes_per_launch_corrected_x2": fb, "write_bytes_per_launch_uncalibrated": wb,
               "hbm_bytes_per_launch": fb + wb,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace on "
                         "PREMVOS_PIPELINE_SERIAL=1 bench.py --steps 1 --warmup 1 with a pre-populated PREMVOS_TUNE_CACHE "
                         "(no autotune trial launches); KB*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                         "half of wide coalesced reads); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted. " + note},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
