"""Dev tool: tools/dev/s8_bench.py's JSON lines -> the table kept as profiles/rNN_s8_bench.txt.  python tools/dev/s8_bench_fmt.py in.jsonl > out.txt"""
import json, sys
print("tools/dev/s8_bench.py on one MI355X (tools/profile_round.sh): csrc/conv_bf16x3_s8.hip per layer shape and tile; TFLOP/s-EQUIVALENT = algorithmic 2 M K N / time")
print("(three bf16 MFMAs per product: issued fraction of the 2.5 PFLOP/s pipe = 3 x that / 2500); fp32 = the kernel the shipped table picks for the same layer (hint 4 = F(4x4,3x3), 5 = streaming).")
print("tiles: 0 = 256x256 / 8 waves / 2 x 64 KB; 1 = 256x128 / 4 waves / 3 buffers; 2 = same / 2 buffers; 3 = 256x128 / 8 waves of 64x64 / 3 buffers; 4, 5 = 128x128 / 3, 2 buffers;")
print("6, 7 = 256x128 on 64-byte rows (3, 2 buffers; two workgroups per CU); 8 = 128x256 on 64-byte rows; 9 = 256x64; 10 = 256x256 ping-pong wave groups (pointwise only);")
print("11 = 256x256 / 4 waves of 128x128.  f32 out / S8 out per tile.\n")
for ln in open(sys.argv[1]):
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    print(f"{d['layer']}   ({d['gflop']} GFLOP)   fp32 {d['fp32_tf']} TFLOP/s {d['fp32_us']} us hint {d['fp32_hint']}   best S8 {d['best_s8_tf']} = "
          f"{d['best_issued_frac_of_bf16_pipe']} issued, x{d['speedup_vs_fp32']}")
    tiles = sorted({int(k[1:].split('_')[0]) for k in d if k.startswith('t') and k.endswith('_tf')})
    cells = [f"t{t}: {d.get(f't{t}_f32out_tf', '-')}/{d.get(f't{t}_s8out_tf', '-')}" for t in tiles]
    err = max([v for k, v in d.items() if k.endswith('_err')] + [0.0])
    print("    " + "  ".join(cells) + f"   max rel. deviation from the fp32 kernel {err:.1e}")
