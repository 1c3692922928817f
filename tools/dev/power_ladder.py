"""Dev tool (round 5): socket power and shader clock of the fp32 GEMM loop's ablation ladder (tools/dev/mfma_ablate.hip `power <v> <s>`):
0 pure MFMA | 1 + ds_read_b128 fragments | 2 + barrier per stage | 3 + ds_write_b128 staging | 4 + global loads | 5 LDS-DMA staging
instead of 3 + 4 | 6 B fragments from global memory.  Which part of the loop draws the 0.5 kW that the pure-MFMA loop does not?"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (BoxSampler)
exe = os.path.join(ROOT, "tools", "dev", "mfma_ablate")
names = ["pure MFMA (operands in registers)", "+ ds_read_b128 fragment reads", "+ one barrier per stage", "+ ds_write_b128 staging (constant data)",
         "+ global loads (L2 / HBM -> registers -> LDS)", "LDS-DMA staging instead (global_load_lds)", "A via LDS, B fragments straight from global memory"]
for v, name in enumerate(names):
    p = subprocess.Popen([exe, "power", str(v), "4"], stdout=subprocess.PIPE, text=True)
    time.sleep(1.0)                                        # past the start-up transient
    with bench.BoxSampler(0, period_s=0.05) as s:
        time.sleep(2.5)
    out = p.communicate()[0].strip()
    b = s.summary()
    clk, pw = (b.get("sclk_mhz_mean_of_xcds") or {}).get("mean"), (b.get("socket_power_w") or {}).get("mean")
    print(f"{v} {name:52s} {out.split(':')[-1].strip():>14s}  {pw} W  {clk} MHz  power-limited {b.get('power_limited_share')}", flush=True)
