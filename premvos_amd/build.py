"""Build recipe for libpremvos_hip.so (hipcc, gfx950 only, in-tree so it travels with gpurun)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PREMVOS_LIB_PATH: developer A/B runs load another build of the library (tools/dev/ab_build.sh); never set in production
LIB = os.environ.get("PREMVOS_LIB_PATH") or os.path.join(CSRC, "libpremvos_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpremvos_hip.so cannot be built")


def sources() -> "list[str]":
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _file_flags(src: str) -> "list[str]":
    """Extra flags a source asks for in a leading ``// hipcc-flags: ...`` comment line."""
    out = []
    with open(src) as f:
        for line in f:
            if not line.startswith("//"):
                break
            if line.startswith("// hipcc-flags:"):
                out += [t for t in line[len("// hipcc-flags:"):].split() if t.startswith("-")]
    return out


def needs_build() -> bool:
    if os.environ.get("PREMVOS_LIB_PATH"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source to objects (parallel) and link the shared library."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
             "-Wno-unused-result"] + os.environ.get("PREMVOS_EXTRA_HIPCC_FLAGS", "").split()      # dev experiments (-D...)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))),
                os.path.getmtime(os.path.join(HERE, "..", "include", "premvos_hip.h"))):
            cmd = [hipcc, *flags, *_file_flags(src), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, verbose=True))
