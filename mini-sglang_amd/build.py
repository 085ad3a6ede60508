"""Build recipe for the gfx950 shared objects (no cmake, no JIT cache).

    python mini-sglang_amd/build.py            # build what is stale
    python mini-sglang_amd/build.py --force

Outputs (git-ignored, shipped to the GPU box by gpurun because they are in-tree):
    mini-sglang_amd/lib/libmsgl_hip.so    every kernel + host helpers   (include/msgl_hip.h)
    mini-sglang_amd/lib/libmsgl_comm.so   RCCL communicator             (links librccl)
    mini-sglang_amd/lib/libmsgl_gemm.so   projection GEMMs + tuner      (links libhipblaslt)

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the build container.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIB = ROOT / "lib"
OBJ = ROOT / "build"
INCLUDE = ROOT.parent / "include"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = os.environ.get("HIPCC", str(ROCM / "bin" / "hipcc"))
ARCH = "gfx950"

HIP_SOURCES = [
    "host.cpp",
    "radix.cpp",
    "store_index.hip",
    "norm_rope_act.hip",
    "attn_decode.hip",
    "attn_prefill.hip",
    "sampling.hip",
    "gemm_skinny.hip",
    "gemm_rowstream.hip",
    "gemm_wstream.hip",
    "gemm_m256.hip",
    "gemm_g3.hip",
    "gemm_ro.hip",
    "comm_p2p.hip",
]
COMM_SOURCES = ["comm.cpp"]
GEMM_SOURCES = ["gemm.cpp"]

COMMON_FLAGS = [
    "-O3",
    "-std=c++17",
    "-fPIC",
    f"--offload-arch={ARCH}",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-inline-asm",
    "-I",
    str(INCLUDE),
]
if os.environ.get("MSGL_PREFILL_DIAG") == "1":  # timing-only ablation variants of the prefill kernels (tools/prefill_ablate.py)
    COMMON_FLAGS.append("-DMSGL_PREFILL_DIAG")


def _stale(out: Path, deps: list[Path]) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps if d.exists())


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + proc.stdout + proc.stderr)
        raise RuntimeError(f"build step failed: {cmd[-1]}")
    if proc.stderr.strip():
        sys.stderr.write(proc.stderr)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ / (src.name + ".o")
    headers = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    if force or _stale(obj, [src] + headers):
        _run([HIPCC, *COMMON_FLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)])
    return obj


def build_all(force: bool = False, verbose: bool = True) -> dict[str, Path]:
    LIB.mkdir(exist_ok=True)
    OBJ.mkdir(exist_ok=True)
    hip_srcs = [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()]
    comm_srcs = [CSRC / s for s in COMM_SOURCES if (CSRC / s).exists()]
    gemm_srcs = [CSRC / s for s in GEMM_SOURCES if (CSRC / s).exists()]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, force), hip_srcs + comm_srcs + gemm_srcs))
    hip_objs = objs[: len(hip_srcs)]
    comm_objs = objs[len(hip_srcs): len(hip_srcs) + len(comm_srcs)]
    gemm_objs = objs[len(hip_srcs) + len(comm_srcs):]
    out = {}
    lib_hip = LIB / "libmsgl_hip.so"
    if force or _stale(lib_hip, hip_objs):
        # --no-undefined: a kernel whose host-side launch stub went missing must fail HERE, not at dlopen on the GPU box
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,--no-undefined", *map(str, hip_objs), "-o", str(lib_hip)])
    out["hip"] = lib_hip
    if comm_objs:
        lib_comm = LIB / "libmsgl_comm.so"
        if force or _stale(lib_comm, comm_objs):
            _run([HIPCC, "-shared", "-fPIC", *map(str, comm_objs), "-L", str(ROCM / "lib"), "-lrccl",
                  f"-Wl,-rpath,{ROCM / 'lib'}", "-o", str(lib_comm)])
        out["comm"] = lib_comm
    if gemm_objs:
        lib_gemm = LIB / "libmsgl_gemm.so"
        if force or _stale(lib_gemm, gemm_objs):
            _run([HIPCC, "-shared", "-fPIC", *map(str, gemm_objs), "-L", str(ROCM / "lib"), "-lhipblaslt",
                  f"-Wl,-rpath,{ROCM / 'lib'}", "-o", str(lib_gemm)])
        out["gemm"] = lib_gemm
    if verbose:
        for k, v in out.items():
            print(f"[build] {k}: {v} ({v.stat().st_size >> 10} KiB)")
    return out


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
