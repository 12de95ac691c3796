"""Build libdnmf_hip.so (gfx950) in-tree with hipcc.

    python -m dnmf_amd.build [--force] [--out PATH] [-DNAME=VALUE | -fFLAG | -mFLAG ...]

``--out`` / ``-D``: a kernel-variant build beside the product library (timing studies; ``DNMF_LIB=PATH`` selects it).

hipcc cross-compiles without a GPU.  The library links only against the HIP runtime; torch is not
involved.  ``-ffp-contract=off``: the coordinate round trip of the warp must not be contracted into FMAs
(see csrc/common.hpp); the kernels spell out ``fmaf`` where a fused multiply-add is wanted.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdnmf_hip.so")
SOURCES = ["api_common.hip", "warp_gather.hip", "recon_image.hip", "warp_recon_grad.hip", "warp_gram_rhs.hip", "warp_gram_sparse.hip", "warp_gram_lists.hip", "recon_lists.hip",
           "mu_temporal.hip", "render_frames.hip", "adam_epoch.hip", "spatial_update.hip", "image_iwarp.hip", "collective.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "dnmf_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


# what the last build_library() call did: "compiled" or "reused" (the library was newer than every source)
LAST_ACTION = None


def build_library(force: bool = False, verbose: bool = True, out: str | None = None, defines=()) -> str:
    """Compile every HIP source into ``dnmf_amd/libdnmf_hip.so`` (or ``out``); returns its path."""
    global LAST_ACTION
    target = out or LIB
    if out is None and not force and not _stale():
        LAST_ACTION = "reused"
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(os.path.abspath(target)), exist_ok=True)
    cmd = [hipcc, *FLAGS, *defines, *[os.path.join(CSRC, s) for s in SOURCES], "-o", target, "-ldl"]
    if verbose:
        print("[dnmf_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    LAST_ACTION = "compiled"
    return target


if __name__ == "__main__":
    args = sys.argv[1:]
    out = args[args.index("--out") + 1] if "--out" in args else None
    print(build_library(force="--force" in args, out=out, defines=[a for a in args if a.startswith(("-D", "-f", "-m"))]))
