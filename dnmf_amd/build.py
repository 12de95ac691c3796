"""Build libdnmf_hip.so (gfx950) in-tree with hipcc.

    python -m dnmf_amd.build [--force] [--out PATH] [--only FILE.hip[,FILE.hip]] [-DNAME=VALUE | -fFLAG | -mFLAG ...]

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
SOURCES = ["api_common.hip", "warp_gather.hip", "recon_image.hip", "warp_recon_grad.hip", "warp_gram_rhs.hip", "warp_gram_sparse.hip", "warp_gram_lists.hip", "warp_gram_lists_z.hip", "recon_lists.hip",
           "mu_temporal.hip", "render_frames.hip", "adam_epoch.hip", "spatial_update.hip", "image_iwarp.hip", "register_patches.hip", "collective.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "dnmf_hip.h")]
    deps.append(os.path.abspath(__file__))      # the per-file flags live here
    return any(os.path.getmtime(d) > t for d in deps)


# sources that include another source
EXTRA_DEPS = {"warp_gram_lists_z.hip": ["warp_gram_lists.hip"]}
# what the last build_library() call did: "compiled" or "reused" (the library was newer than every source)
LAST_ACTION = None
# flags of single files (measured per kernel)
PER_FILE_FLAGS = {
    # K2: the SLP vectoriser's packed fp32 operations issue no faster than two scalar ones and cost register shuffles
    # (512x512x2x4000: 6.55 ms against 6.93; Z == 1 unchanged)
    "warp_recon_grad.hip": ["-fno-slp-vectorize"],
    # K7: the same -- the packed operations come with s_nop wait states between dependent ones
    # (512x512 x 4000 frames: 3.65 ms against 4.14; 512x512x2 x 1000: 5.56 against 7.01)
    "image_iwarp.hip": ["-fno-slp-vectorize"],
    # K3n for Z >= 2: packed operations want their operands in register pairs -- 200 registers instead of 150, spills
    # (512x512x2x4000: 19.9 ms against 6.6)
    "warp_gram_lists_z.hip": ["-fno-slp-vectorize"],
    # ... and its Z == 1 instantiations: 2.95 ms against 3.01 per 4000 frames of 512x512, K = 100, same box, on the final
    # code of round 3 (an earlier state of the kernel spilled vector registers without the vectoriser and lost)
    "warp_gram_lists.hip": ["-fno-slp-vectorize"],
}


def source_hashes() -> dict:
    """{file name: first 12 hex digits of its sha256} for every file under csrc/ and the public header."""
    import hashlib
    files = {f: os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))}
    files["dnmf_hip.h"] = os.path.join(HERE, "..", "include", "dnmf_hip.h")
    return {name: hashlib.sha256(open(path, "rb").read()).hexdigest()[:12] for name, path in files.items()}


def _compile_one(job):
    cmd, src = job
    subprocess.run(cmd, check=True)
    return src


def build_library(force: bool = False, verbose: bool = True, out: str | None = None, defines=(), only=None) -> str:
    """Compile every HIP source into ``dnmf_amd/libdnmf_hip.so`` (or ``out``); returns its path.

    One object per source under ``build/obj/<key>/`` (key = the extra flags), compiled in parallel and reused while it is
    newer than its source and every header: an edit to one kernel file recompiles that file only.  ``only``: source
    names the extra ``defines`` apply to (a variant of one kernel file links against the product objects of the others)."""
    global LAST_ACTION
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    target = out or LIB
    if out is None and not force and not _stale():
        LAST_ACTION = "reused"
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(os.path.abspath(target)), exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]

    def objdir_for(extra):
        key = hashlib.sha1(" ".join([hipcc, *cflags, *extra]).encode()).hexdigest()[:12]
        d = os.path.join(HERE, "..", "build", "obj", key)
        os.makedirs(d, exist_ok=True)
        return d
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "dnmf_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    jobs, objs = [], []
    stamp = "-DDNMF_BUILD_STAMP=\"" + ";".join(f"{k}:{v}" for k, v in sorted(source_hashes().items())) + "\""
    for s in SOURCES:
        extra = [*PER_FILE_FLAGS.get(s, []), *(defines if (only is None or s in only) else [])]
        if s == "api_common.hip":
            extra.append(stamp)
        src, obj = os.path.join(CSRC, s), os.path.join(objdir_for(extra), s + ".o")
        objs.append(obj)
        dep_time = max([os.path.getmtime(src), hdr_time, *[os.path.getmtime(os.path.join(CSRC, d)) for d in EXTRA_DEPS.get(s, [])]])
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < dep_time:
            jobs.append(([hipcc, *cflags, *extra, "-c", src, "-o", obj], s))
    if verbose:
        print(f"[dnmf_amd.build] {hipcc} {' '.join([*cflags, *defines])}: compiling {[j[1] for j in jobs]}, "
              f"reusing {len(objs) - len(jobs)} objects", flush=True)
    workers = max(1, min(len(jobs), (os.cpu_count() or 2)))
    if jobs:
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(_compile_one, jobs))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", target, "-ldl"], check=True)
    LAST_ACTION = "compiled"
    return target


if __name__ == "__main__":
    args = sys.argv[1:]
    out = args[args.index("--out") + 1] if "--out" in args else None
    only = args[args.index("--only") + 1].split(",") if "--only" in args else None
    print(build_library(force="--force" in args, out=out, defines=[a for a in args if a.startswith(("-D", "-f", "-m"))],
                        only=only))
