"""Dev tool: file-to-file rate of the streaming driver (bench.file_to_file) over chunk size / refinement group / lanes / GPU JPEG."""
import json, os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for chunk, group, lanes, gj in ((8, 4, 2, 0), (8, 4, 2, 1), (16, 8, 2, 1), (8, 8, 2, 1), (16, 4, 2, 1), (8, 4, 1, 1), (8, 4, 3, 1), (4, 4, 2, 1)):
    os.environ.update(PREMVOS_DRIVER_BATCH=str(group), PREMVOS_STREAM_REFINE_LANES=str(lanes), PREMVOS_GPU_JPEG=str(gj))
    r = bench.file_to_file(frames, chunk)
    print(json.dumps({"chunk": chunk, "refine_group": group, "refine_lanes": lanes, "gpu_jpeg": gj, "fps": r["streaming_driver_fps"],
                      "warm": r["warm_runs_s"], "cold": r["cold_run_s"], "props": r["proposals_per_frame"]}), flush=True)
    gc.collect(); torch.cuda.empty_cache()
