#!/usr/bin/env python3
"""Headline benchmark: frames/sec demixed on the WUtils.Simulator video, 512x512xT, K=100.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" is one full sweep of the fit over the resident video, the unit the reference's demo repeats
(demo.py:44-46): ``update_motion(epochs=1)`` (mini-batches of 4, Adam lr 1e-5 on beta) followed by
``update_footprints(iter_c=50, gamma_c=0)``.  N=1 workload = BASELINE.json configs[2]: 512x512 (Z=1),
T=4000, K=100, fp32.  N>1: the T axis is sharded, every rank holds 4000 frames of one 4000*N-frame video
(weak scaling); there is no data-path collective (frames are independent in both steps), only the timing
barrier.  The video is generated on the GPU before the timed region starts (inputs resident in HBM).

``--with-spatial`` adds the footprint update the reference leaves commented out (Demix/dNMF.py:169-176) to every
sweep: K7 registration, K5, ONE all-reduce of the A1 | C_s buffer over the ranks (RCCL for N > 1) and K6 -- the only
collective the path has; it is a separate mode, never the headline value.

Rank 0 prints ONE JSON line; it also carries
  roofline      the dominant kernel (the Gram kernel K3n / K3s / K3) timed with HIP events on its stream
  cpu_baseline  the CPU oracle (reference op sequence) timed on this host on a bounded sample, N=1 only
  ranks         (N > 1) world size, backend and the (rank, device, PCI bus) rows all-gathered over the process group
  extras        (default N=1 run only, outside the timed region, a few sweeps each; --no-extras skips them)
                depth2: the same sweep at 512x512x2 -- the smallest volume the reference itself can run, 8-tap kernels;
                with_spatial: the sweep of --with-spatial with its kernel times;
                stock_dataloader: the sweep driven by a plain torch DataLoader over a HOST video as in demo.py:33-35
                (configs[1] whole, a 400-frame subset of configs[2]): PCIe- and host-inclusive frames/s
"""
import argparse
import gc
import json
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.utils.data

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak (spec)
HBM_PEAK_GBS = 8000.0         # same guide: HBM3E ~8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a sweep takes ~10 ms; the first few after an idle period run ~30 % slower (clock ramp), hence the defaults
    # 20 + 5 sweeps: the fit stays where the demo's would be after as many (Adam's momentum moves every coefficient of
    # every frame at every step, dNMF.py:191; after ~100 sweeps the warps have drifted by tens of voxels and the traces of
    # neurons that lost their support grow without bound -- finite, but no longer the workload); `sweep_ms_on_stream`
    # shows how even the timed sweeps were
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512, help="X = Y (Z = 1)")
    ap.add_argument("--frames", type=int, default=4000, help="frames per GPU")
    ap.add_argument("--neurons", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--lr", type=float, default=None,
                    help="Adam learning rate on beta; default 1e-5 * (50 / size)^2: the demo's 1e-5 (demo.py:42) moves the "
                         "far corner of its 50 x 50 volume by 0.025 px per step through the quadratic coefficients; the same "
                         "number on a 512 x 512 volume is 2.6 px per step, and the fit (the reference's too) throws whole "
                         "footprints out of the volume within one epoch, after which traces overflow and beta turns NaN")
    ap.add_argument("--iter-c", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--gram", choices=["auto", "dense", "sparse", "lists"], default="auto",
                    help="Gram kernel: K3 (dense), K3s (exact-zero blocks skipped), K3n (neuron lists) or by footprint shape")
    ap.add_argument("--depth", type=int, default=1, help="Z (slices); the headline workload is Z = 1")
    ap.add_argument("--with-spatial", action="store_true",
                    help="sweep = default + update_footprints(live_spatial=True): K7 registration, K5, ONE all-reduce of "
                         "A1 | C_s over the ranks (RCCL for N > 1), K6 -- the path's only collective")
    ap.add_argument("--exchange-extra", action="store_true",
                    help="N > 1 only: after the timed region also run the live footprint update (the path's one all-reduce) "
                         "twice under a watchdog and report it in extras.sharded_footprint_update; a rank that fails or a "
                         "collective that does not return makes every rank leave with exit code 3 after rank 0 has printed "
                         "the line with the error.  Off by default: the default --gpus N line has no post-region hazard "
                         "(--with-spatial times the collective as the main line)")
    ap.add_argument("--beta-init", choices=["identity", "displaced"], default="identity",
                    help="displaced: every frame starts from a smooth warp about a voxel away from the identity (shift, linear "
                         "and quadratic terms) instead of the identity")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements of the default N=1 run (Z = 2 line, spatial mode, stock DataLoader)")
    return ap.parse_args()


class Traffic:
    """profiles/roofline_traffic.json: HBM bytes and issue statistics per launch from separate rocprofv3 --pmc passes
    (tools/make_traffic_json.py).  Every entry names the source files its kernel is built from and the file carries their
    hashes at measurement time; an entry is handed out only when those hashes equal the ones compiled into the library
    this process loaded (dnmf_build_stamp), else None -- a counter file cannot go stale silently."""

    def __init__(self, lib_stamp):
        self.entries, self.lib = {}, lib_stamp
        path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(path):
            self.entries = json.load(open(path)).get("entries", {})
        self.stale = set()

    def get(self, key):
        e = self.entries.get(key)
        if e is None:
            return None
        src = e.get("sources", {})
        if not src or any(h is None or h != self.lib.get(f) for f, h in src.items()):
            self.stale.add(key)
            return None
        return e["value"]

    def note(self):
        if not self.stale:
            return None
        return ("entries of profiles/roofline_traffic.json measured on other sources than the loaded library and therefore "
                "reported as null: " + ", ".join(sorted(self.stale)))


def other_kernels(evs, P, K, T_loc, tjson, key, Pp):
    """HBM rooflines of the two other kernels of a sweep that move data, from the same HIP-event hooks.  Pp = floats of
    a reconstruction image in the halo layout (what is actually written / addressed)."""
    out = []
    mg = evs("motion_grad_lists")
    if mg:
        ms = 1e3 * sum(mg) / len(mg)
        b = 12.0 * P * T_loc   # the frame read, the reconstruction image written and read back
        out.append({"kernel": "dnmf_motion_grad_lists: recon_lists_kernel + warp_recon_grad_kernel (K2) + its finish kernel, "
                              "alternating over pieces of frames whose reconstruction images share one cache-resident buffer",
                    "bound": "hbm", "launch_ms": ms, "bytes_per_launch": b, "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": b / ms / 1e6 / HBM_PEAK_GBS,
                    "count": "algorithmic bytes 12PT = frame 4PT + reconstruction image written 4PT and gathered 4PT; the "
                             "image traffic is meant to stay in the Infinity Cache, so HBM should see ~4PT (traffic)",
                    "traffic": tjson.get(f"{key.split('_K')[0]}_motion")})
    k2 = evs("warp_recon_grad")
    if k2:
        ms = 1e3 * sum(k2) / len(k2)
        b = 4.0 * P * T_loc + 4.0 * Pp * T_loc   # the frame and its reconstruction image (halo layout), each read once
        out.append(hbm_roof("warp_recon_grad_kernel (K2: warp of the reconstruction image, residual, gradient sums)", ms, b,
                            "algorithmic bytes 4PT (frames) + 4 halo(P) T (reconstruction images)",
                            tjson.get(f"{key.split('_K')[0]}_K2")))
    rl = evs("recon_image_lists")
    if rl:
        ms = 1e3 * sum(rl) / len(rl)
        b = 4.0 * P * T_loc   # the reconstruction image, written once (algorithmic; the halo layout adds Pp/P - 1; in steady
        #                       state the tiles without a neuron -- ~10 % here -- are not rewritten, see `traffic`)
        out.append({"kernel": "recon_lists_kernel (reconstruction image from neuron lists)", "bound": "hbm",
                    "launch_ms": ms, "bytes_per_launch": b, "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": b / ms / 1e6 / HBM_PEAK_GBS,
                    "traffic": tjson.get(f"{key.split('_K')[0]}_recon_lists")})
    return out


def usable_cpus():
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(size, K, batch, iter_c, positions0, frames_host, lr):
    """The oracle's faithful restatement of the reference sweep on a bounded sample of the same workload."""
    from oracle import dnmf_oracle as O
    sz = [size, size, 1]
    nb = batch          # frames in the motion sample (one mini-batch)
    nf = 2              # frames in the footprint sample
    torch.manual_seed(0)
    m = O.OracleModel(sz, K, nb, positions0, C0=torch.rand(K, nb).numpy())
    video = np.ascontiguousarray(np.moveaxis(frames_host[:nb].reshape(nb, size, size, 1), 0, 3))
    opt = torch.optim.Adam([m.beta_param], lr=lr)
    t0 = time.perf_counter()
    m.update_motion(video, [list(range(nb))], opt, gamma=1, epochs=1)
    t_motion = (time.perf_counter() - t0) / nb
    t0 = time.perf_counter()
    A_t, _, Yv = m.pushforward(video[..., :nf], nf)
    A_t, Yv = A_t[..., :nf], Yv[..., :nf]
    t_push = (time.perf_counter() - t0) / nf
    C = m.C[:, :nf].astype(np.float64)
    t0 = time.perf_counter()
    O.update_temporal(A_t, C, Yv, gamma=0)
    t_iter = (time.perf_counter() - t0) / nf
    faithful = 1.0 / (t_motion + t_push + iter_c * t_iter)
    hoisted = 1.0 / (t_motion + t_push + t_iter)
    # SURVEY 8(d)'s "optimised CPU" variant: the contraction once, as an fp32 torch.bmm on all usable threads, then the
    # iter_c multiplicative rounds on the K x K data
    At = torch.from_numpy(np.ascontiguousarray(A_t.reshape(-1, K, nf).transpose(2, 0, 1))).float()   # (nf, P, K)
    Yt = torch.from_numpy(np.ascontiguousarray(Yv.reshape(-1, nf).T)).float()                          # (nf, P)
    t0 = time.perf_counter()
    G = torch.bmm(At.transpose(1, 2), At)
    r = torch.bmm(At.transpose(1, 2), Yt[:, :, None])[:, :, 0]
    O.mu_temporal_from_gram(G.permute(1, 2, 0).double().numpy(), r.T.double().numpy(), C, None, iter_c)
    t_opt = (time.perf_counter() - t0) / nf
    optimised = 1.0 / (t_motion + t_push + t_opt)
    return {
        "value": faithful, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"oracle/dnmf_oracle.py at {size}x{size}x1 K={K}: update_motion on one mini-batch of {nb} frames "
                   f"({t_motion:.3f} s/frame, torch-CPU {torch.get_num_threads()} threads), pushforward of {nf} frames "
                   f"({t_push:.3f} s/frame), ONE of the {iter_c} identical update_temporal rounds on those {nf} frames "
                   f"({t_iter:.3f} s/frame/round, numpy einsum, 1 thread) scaled x{iter_c}; every cost is linear in T"),
        "hoisted_value": hoisted,
        "hoisted_note": "same sample with the Gram/rhs contraction done once instead of iter_c times",
        "optimised_value": optimised,
        "optimised_note": (f"same sample with the contraction as one fp32 torch.bmm on {torch.get_num_threads()} threads "
                           f"({t_opt:.3f} s/frame incl. the {iter_c} rounds on the K x K data); the warp steps are "
                           "torch-CPU grid_sample either way"),
    }


class HostVideo(torch.utils.data.Dataset):
    """A host copy of the video behind the reference's dataset protocol (Demix/dNMF.py:196-217): frame-major memory,
    the (X,Y,Z,T) shape as a view, __getitem__ -> (frame, index) with the in-place clamp."""

    def __init__(self, frames, sz):
        self.video = frames.view(frames.shape[0], *sz).permute(1, 2, 3, 0)

    def __len__(self):
        return self.video.shape[3]

    def __getitem__(self, idx):
        sample = self.video[:, :, :, idx]
        sample[sample < 0] = 0
        return sample, idx


def position_initialiser_extra(size, Z=2, T=256):
    """K8: the registration behind MotionCorrect on a size x size x Z video of T frames -- the rigid pass
    (dnmf_rigid_correct: shifts, every frame moved through its spectrum) and the piecewise pass (dnmf_register_patches: 5 x 5
    patches) -- with the roofline of the axis transform: f32 MFMAs.  The flops counted are those of the full-length
    transforms only (forward transforms of the volume and of the patches, the full inverse of the rigid pass: 8 flops per
    complex multiply-add, N (n0 + n1 + n2) of them per box); the windowed and upsampled inverses come on top, uncounted."""
    import time
    from dnmf_amd import ops
    sz = [size, size, Z]
    P = size * size * Z
    gen = torch.Generator(device="cuda").manual_seed(5)
    frames = torch.rand(T, P, device="cuda", generator=gen)
    tmpl = frames.mean(0)
    stride, overlap = (3 * size) // 16, size // 16
    st, ov, ms = (stride, stride, 1), (overlap, overlap, Z - 1), (6, 6, 1 if Z > 1 else 0)
    dims, _ = ops.patch_grid(sz, st, ov)
    w = [a + b for a, b in zip(st, ov)]
    NP = int(dims[0] * dims[1] * dims[2])
    fwd_full = 8.0 * P * (size + size + Z)
    fwd_patch = 8.0 * NP * w[0] * w[1] * w[2] * (w[0] + w[1] + w[2])

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    t_pw = timed(lambda: ops.register_patches(frames, tmpl, sz, st, ov, ms, 3, 10, 0.0))
    t_rg = timed(lambda: ops.rigid_correct(frames, tmpl, sz, ms, 10, 0.0, True))
    peak = 157.3
    out = {"workload": f"{size}x{size}x{Z}, {T} frames, {NP} patches of {w[0]}x{w[1]}x{w[2]}, max_shifts {ms}, upsample factor 10",
           "register_patches_frames_per_s": T / t_pw, "rigid_correct_frames_per_s": T / t_rg,
           "rooflines": [
               {"kernel": "mc_axis_mfma_kernel in dnmf_register_patches (v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                "achieved": T * (fwd_full + fwd_patch) / t_pw / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": T * (fwd_full + fwd_patch) / t_pw / 1e12 / peak, "traffic": None,
                "count": "forward transforms of the volume and of every patch; wall time of the whole call"},
               {"kernel": "mc_axis_mfma_kernel in dnmf_rigid_correct", "bound": "mfma", "achieved": T * 2 * fwd_full / t_rg / 1e12,
                "peak": peak, "unit": "TFLOP/s", "frac": T * 2 * fwd_full / t_rg / 1e12 / peak, "traffic": None,
                "count": "forward transform of the volume + the full inverse on the shifted grid; wall time of the whole call"}],
           "note": "parity unpinned (DESIGN 2: the reference module cannot be imported here and holds no fixture)"}
    del frames
    torch.cuda.empty_cache()
    return out


def displaced_beta(T, sz, device, seed=7):
    """(10,3,T) warps about a voxel off the identity: shifts of +-1 px, linear terms of +-0.2 %, quadratic terms that move
    the far corner by ~1 px, z row left at the identity; smooth in t (random walk of period ~200 frames).  (Warps several
    voxels away from where the simulator put the neurons are not a fit any more: a trace whose footprint has lost its
    neuron grows without bound under the multiplicative update, and fit_sanity rejects the run.)"""
    g = torch.Generator().manual_seed(seed)
    S = float(max(sz[0], sz[1]))
    amp = torch.tensor([1.0] + [0.002] * 3 + [1.0 / S ** 2] * 6)[:, None, None]         # per basis term
    knots = torch.randn(10, 3, max(2, T // 200 + 2), generator=g)
    walk = torch.nn.functional.interpolate(knots.reshape(1, 30, -1), size=T, mode="linear", align_corners=True).reshape(10, 3, T)
    d = amp * walk
    d[:, 2, :] = 0.0
    d[[3, 6, 8, 9], :, :] = 0.0     # terms with z
    ident = torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None]
    return (ident + d).to(device).contiguous()


def run_sweeps(args, sz, K, T_loc, steps, warmup, rank, world, group, with_spatial=False, loader="resident",
               beta_init="identity", footprint_floor=0.0):
    """Build the workload (synthetic video resident in HBM, model, loaders) and time `steps` sweeps after `warmup`.
    loader = "resident": the fit reads the rows where they lie; "dataloader": a stock torch DataLoader over a host
    copy of the video behind a plain Dataset (demo.py:33-35), every sweep crosses PCIe twice; "dataset": a stock
    DataLoader over the library's SimulatedVideoDataset holding that host copy.  Returns a dict of measurements."""
    from dnmf_amd import ops
    from dnmf_amd.Demix import dNMF as M
    from dnmf_amd.WUtils import Simulator
    bs = args.batch
    T_total = T_loc * world
    torch.manual_seed(0)
    np.random.seed(0)
    par = {"sigma": [5, 5, .01], "ls": [10, 10, 10]}
    frames, positions, _ = Simulator.generate_video_resident(K, T_total, sz, 3, .2, -120, par, t0=rank * T_loc,
                                                             t1=(rank + 1) * T_loc, group=group)
    frames.clamp_(min=0)  # what the dataset's __getitem__ does to every frame it serves (dNMF.py:214-215)
    positions0 = positions[:, :, 0].contiguous()
    torch.manual_seed(1 + rank)
    dn = M.DeformableNMF(torch.tensor(sz), K, T_loc, positions=positions0)
    dn.verbose = False
    dn.gram_kernel = args.gram
    dn.group = group
    dn.fp.footprint_floor = footprint_floor
    if beta_init == "displaced":
        with torch.no_grad():
            dn.fp.beta.copy_(displaced_beta(T_loc, sz, dn.fp.beta.device))
    beta_start = dn.fp.beta.detach().clone()
    lr = args.lr if args.lr is not None else 1e-5 * (50.0 / sz[0]) ** 2
    opt = torch.optim.Adam([dn.fp.beta], lr=lr)
    if loader == "resident":
        # every rank draws the SAME global mini-batch order and keeps its own frames: the optimiser-step sequence
        # is the single-process one for the 4000*N-frame video (dnmf_amd/sharding.py)
        gen = torch.Generator().manual_seed(1234)
        train = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=gen, t0=rank * T_loc, T_total=T_total)
        test = M.ResidentLoader(frames, sz, bs, shuffle=False)
    else:
        if loader == "dataset":
            # the library's own dataset class around a host video, as demo.py builds it (its constructor would run the CPU
            # simulator: hours at this size): a stock DataLoader over it is served from dataset.device_frames()
            host = M.SimulatedVideoDataset.__new__(M.SimulatedVideoDataset)
            host.video = frames.cpu().view(frames.shape[0], *sz).permute(1, 2, 3, 0)
            host.positions, host.traces, host._sz = positions, None, list(sz)
        else:
            host = HostVideo(frames.cpu(), sz)
        del frames
        frames = None
        train = torch.utils.data.DataLoader(host, batch_size=bs, shuffle=True, generator=torch.Generator().manual_seed(1234))
        test = torch.utils.data.DataLoader(host, batch_size=bs, shuffle=False)

    def step():
        dn.update_motion(train, opt, gamma=1, epochs=1)
        dn.update_footprints(test, bs, sz, gamma_c=0, gamma_a=1.0, iter_c=args.iter_c, return_dense=False,
                             live_spatial=with_spatial)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ops.TIMING = {}           # hooks on during warm-up too: their first use has one-time costs (event pool, counters)
    for _ in range(max(1, warmup)):
        step()
    fence()
    ops.TIMING = {}
    for counters in (ops.SPARSE_COUNTERS, ops.LISTS_COUNTERS):
        if counters is not None:
            counters.zero_()
    fence()
    # one event per sweep on the stream the kernels run on: the spread of the sweep times goes into the line
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    gc.collect()
    gc.disable()   # no collector pauses between launches inside the timed region
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    sweep_each = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    slowest = max(range(steps), key=lambda i: sweep_each[i])
    sweep_ms = sorted(sweep_each)
    timing, ops.TIMING = ops.TIMING, None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt[0])
    lists_counters = None if ops.LISTS_COUNTERS is None else ops.LISTS_COUNTERS.tolist()
    sparse_counters = None if ops.SPARSE_COUNTERS is None else ops.SPARSE_COUNTERS.tolist()
    # the sweeps must have worked on a sane fit: a frame whose beta is not finite gathers nothing and costs K3n nothing
    beta = dn.fp.beta.detach()
    ident = torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].to(beta.device)
    sanity = {"finite_beta_frames": int(torch.isfinite(beta).all(0).all(0).sum()), "frames": T_loc,
              "beta_init": beta_init, "max_abs_beta_minus_start": float((beta - beta_start).abs().nan_to_num(float("inf")).max()),
              "finite_trace_entries": int(torch.isfinite(dn.C).sum()), "trace_entries": dn.C.numel(),
              "max_abs_beta_minus_identity": float((beta - ident).abs().nan_to_num(float("inf")).max()),
              "max_trace": float(dn.C.nan_to_num(float("inf")).max()), "lr": lr}
    # (a trace far above the video's range -- the simulator's frames peak near 1 -- means a neuron lost its support)
    sanity["ok"] = (sanity["finite_beta_frames"] == T_loc and sanity["finite_trace_entries"] == dn.C.numel()
                    and sanity["max_trace"] < 1e3)
    if not sanity["ok"]:
        print(f"[bench] WARNING: the fit left the sane range: {sanity}", file=sys.stderr, flush=True)

    def evs(name):
        return [a.elapsed_time(b) * 1e-3 for a, b in timing.get(name, [])]

    per_step = {name: 1e3 * sum(evs(name)) / steps for name in timing}
    return {"elapsed": elapsed, "evs": evs, "per_step_ms": per_step, "dn": dn, "frames": frames, "positions0": positions0,
            "sanity": sanity, "lr": lr,
            "sweep_ms": {"min": round(sweep_ms[0], 3), "median": round(sweep_ms[len(sweep_ms) // 2], 3),
                         "max": round(sweep_ms[-1], 3), "slowest_is_sweep": slowest},
            "lists_counters": lists_counters, "sparse_counters": sparse_counters, "T_total": T_total}


def sharded_footprint_update(dn, frames, sz, bs, iter_c, group, world, updates=2):
    """N > 1 only, after the timed region: the one step of the path that exchanges data between ranks -- the live
    footprint update (registration K7, K5 over the local frames, ONE all-reduce of the A1 | C_s buffer over RCCL, K6).
    Every rank must end with the same footprints."""
    import torch.distributed as dist
    from dnmf_amd import ops
    from dnmf_amd.Demix import dNMF as M
    test = M.ResidentLoader(frames, sz, bs, shuffle=False)
    P, K = dn.fp.P, dn.fp.K

    def update():
        dn.update_footprints(test, bs, sz, gamma_c=0, gamma_a=1.0, iter_c=iter_c, return_dense=False, live_spatial=True)

    update()   # first use: buffers, the registered video
    torch.cuda.synchronize()
    dist.barrier(group=group)
    ops.TIMING = {}
    t0 = time.perf_counter()
    for _ in range(updates):
        update()
    torch.cuda.synchronize()
    dist.barrier(group=group)
    elapsed = time.perf_counter() - t0
    timing, ops.TIMING = ops.TIMING, None
    ar = [a.elapsed_time(b) for a, b in timing.get("allreduce", [])]
    check = dn.fp.A.double().sum().reshape(1)
    sums = [torch.zeros_like(check) for _ in range(world)]
    dist.all_gather(sums, check, group=group)
    nbytes = 4 * dn._spatial_buf.numel()   # A1c | C_s: the compact buffer when the footprints are compact, else P K + K^2 floats
    return {"what": "update_footprints(live_spatial=True) on every rank: K3n + K4, K7, K5, one all-reduce (sum) of A1 | C_s, K6",
            "backend": "torch.distributed 'nccl' = RCCL" if dist.get_backend(group) == "nccl" else f"torch.distributed '{dist.get_backend(group)}'",
            "ranks": world,
            "ms_per_update_max_over_ranks": 1e3 * elapsed / updates,
            "allreduce_ms_rank0": round(sum(ar) / max(1, len(ar)), 3), "allreduce_calls": len(ar), "allreduce_bytes": nbytes,
            "allreduce_algbw_GBps": round(nbytes / (1e6 * max(1e-9, sum(ar) / max(1, len(ar)))), 1),
            "footprints_identical_on_all_ranks": bool(all(float(v) == float(sums[0]) for v in sums))}


def checked_value(res, steps):
    """frames/s of a run, or (None, the rate) when its fit left the sane range: such a run is not a measurement of the
    workload (a frame whose beta is not finite costs the kernels nothing)."""
    v = res["T_total"] * steps / res["elapsed"]
    return (v, None) if res["sanity"]["ok"] else (None, v)


def hbm_roof(kernel, ms, nbytes, count, traffic=None):
    return {"kernel": kernel, "bound": "hbm", "launch_ms": ms, "bytes_per_launch": nbytes, "achieved": nbytes / ms / 1e6,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS, "count": count,
            "traffic": traffic}


def short_line(res, steps, rooflines=None):
    """frames/s, ms per sweep and the per-kernel HIP-event times of a secondary measurement."""
    v, bad = checked_value(res, steps)
    out = {"value": v, "unit": "frames/s", "steps": steps,
           "ms_per_step": 1e3 * res["elapsed"] / steps, "sweep_ms_on_stream": res["sweep_ms"], "fit_sanity": res["sanity"],
           "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(res["per_step_ms"].items())}}
    if bad is not None:
        out["value_of_the_invalid_run"] = bad
    if res.get("lists_counters"):
        n = max(1, len(res["evs"]("warp_gram_rhs_lists")))
        out["tile_neuron_evaluations_per_frame"] = res["lists_counters"][0] / n / (res["T_total"])
    if rooflines:
        out["rooflines"] = rooflines(res)
    return out


def main():
    args = parse()
    # torch sizes its CPU thread pool by the host's core count; under a smaller cgroup quota every CPU op then gets
    # the process throttled for most of a 100 ms scheduler period -- longer than a sweep takes on the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // world)))  # the ranks share the quota
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    group = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
        group = dist.group.WORLD

    from dnmf_amd import ops

    size, K, T_loc, bs, Z = args.size, args.neurons, args.frames, args.batch, args.depth
    sz = [size, size, Z]
    P = size * size * Z
    T_total = T_loc * world
    ntap = 8 if Z > 1 else 4

    res = run_sweeps(args, sz, K, T_loc, args.steps, args.warmup, rank, world, group, with_spatial=args.with_spatial,
                     beta_init=args.beta_init)
    elapsed, evs, dn, frames, positions0 = res["elapsed"], res["evs"], res["dn"], res["frames"], res["positions0"]

    # one launch of the dense Gram kernel outside the timed region, for the roofline of the kernel that
    # evaluates every product (the timed sweeps may have used the zero-skipping kernel instead)
    dense_ms = None
    if rank == 0 and K <= 127 and not args.no_extras:  # one K3 launch holds at most 127 neurons
        torch.cuda.synchronize()
        ops.TIMING = {}
        ops.warp_gram_rhs(dn.fp.packed_footprints(), K, sz, dn.fp.beta.detach(), None, frames)
        torch.cuda.synchronize()
        (a, b), = ops.TIMING["warp_gram_rhs"]
        dense_ms, ops.TIMING = a.elapsed_time(b), None

    line = None
    if rank == 0:
        k3d, k3s, k3n, k2 = (evs("warp_gram_rhs"), evs("warp_gram_rhs_sparse"), evs("warp_gram_rhs_lists"),
                             evs("warp_recon_grad"))
        lists, sparse = len(k3n) > 0, len(k3s) > 0
        k3 = k3n if lists else (k3s if sparse else k3d)
        k3_avg = sum(k3) / max(1, len(k3))
        # algorithmic flops of one Gram launch when every product is evaluated (symmetric Gram,
        # rhs, 4 or 8 interpolation taps)
        dense_flops = T_loc * (P * K * (K + 1) + 2 * P * K + 2 * ntap * P * K)
        tjson = Traffic(ops.build_stamp())
        key = f"{size}x{size}x{T_loc}_K{K}" if Z == 1 else f"{size}x{size}x{Z}x{T_loc}_K{K}"
        if lists:
            # K3n has no matrix-pipe work; its floor is the traffic it cannot avoid: every frame once, the footprints
            # once (they stay in L2 / MALL across frames), G and r once
            abytes = 4.0 * P * T_loc + 4.0 * P * K + 4.0 * T_loc * (K * K + K)
            n_eval, n_pair = (float(v) / len(k3n) for v in res["lists_counters"][:2])
            roof = {"kernel": f"warp_gram_lists_kernel<{ntap},{1 if K <= 64 else 2 if K <= 128 else 4},1,true,*> (K3n: per "
                              "256-voxel tile only the neurons whose non-zero box the tile's taps can reach; vector ALU, no "
                              "MFMA; two launches: tiles with up to four neurons, then the others) + lists_tilemask_kernel "
                              "(the tiles' neuron lists), one API call",
                    "bound": "hbm", "achieved": abytes / k3_avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": abytes / k3_avg / 1e9 / HBM_PEAK_GBS, "traffic": tjson.get(key + "_lists"),
                    "launch_ms": 1e3 * k3_avg, "launches": len(k3n), "bytes_per_launch": abytes,
                    "count": "algorithmic bytes = frames 4PT + footprints 4PK (once per launch) + G, r 4T(K^2+K); the "
                             "kernel is bound by vector-ALU issue (the reference's fp32 coordinate sequence once per voxel "
                             "and frame, then gathers and reductions per listed neuron), see valu_issue",
                    "valu_issue": tjson.get(key + "_lists_valu"),
                    "tile_neuron_evaluations_per_launch": n_eval, "tile_pair_sums_per_launch": n_pair,
                    "dense_equivalent_flops_per_launch": dense_flops,
                    "dense_equivalent_tflops": dense_flops / k3_avg / 1e12}
        elif sparse:
            # flops of the products that were not skipped, from the kernel's own counters: an MFMA is
            # 16x16x4 MACs; a (block, k-step) gather is 64 lanes x (taps + rhs) FMAs
            n_mfma, n_blend = (float(v) / len(k3s) for v in res["sparse_counters"][:2])
            flops = n_mfma * 2048 + n_blend * 64 * 2 * (ntap + 1) + T_loc * P * 82  # + warp geometry (82 flops/voxel)
            kname = "warp_gram_lt_kernel" if ops.SPARSE_VARIANT == "table" else "warp_gram_sparse_kernel"
            roof = {"kernel": kname + " (K3s, v_mfma_f32_16x16x4_f32, exact-zero blocks skipped)",
                    "bound": "mfma", "achieved": flops / k3_avg / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / k3_avg / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": tjson.get(key + "_sparse"),
                    "launch_ms": 1e3 * k3_avg, "launches": len(k3s), "flops_per_launch": flops,
                    "count": "flops of the products evaluated (16-neuron blocks that are non-zero in a half pass = 8x4 "
                             "voxels; MFMA + gather + rhs, from in-kernel counters) + 82 flops/voxel of warp geometry; "
                             "products with an exact zero are skipped, so this kernel is bound by instruction issue of "
                             "the per-voxel bookkeeping, not by the matrix pipe",
                    "dense_equivalent_flops_per_launch": dense_flops,
                    "dense_equivalent_tflops": dense_flops / k3_avg / 1e12}
        else:
            roof = {"kernel": "warp_gram_kernel (K3, v_mfma_f32_16x16x4_f32)", "bound": "mfma",
                    "achieved": dense_flops / k3_avg / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": dense_flops / k3_avg / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": tjson.get(key),
                    "launch_ms": 1e3 * k3_avg, "launches": len(k3), "flops_per_launch": dense_flops,
                    "count": "P*K*(K+1) symmetric Gram + 2PK rhs + 2*taps*PK interpolation, per frame"}
        line = {
            "metric": "frames/sec demixed, 512x512xT K=100",
            "value": checked_value(res, args.steps)[0],
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Simulator {size}x{size}x{T_total} (Z={Z}), K={K}, fp32: update_motion(epochs=1, "
                                   f"batch {bs}, Adam lr {res['lr']:.3g}) + update_footprints(iter_c={args.iter_c}, gamma_c=0"
                                   + (", live_spatial=True: K7 + K5 + all-reduce + K6)" if args.with_spatial else ")"),
                       "frames_per_gpu": T_loc,
                       "parallelism": f"frames sharded over {world} GPU(s), " +
                                      ("one RCCL all-reduce of A1 | C_s per sweep" if args.with_spatial else "no collective"),
                       "gram_kernel": dn.gram_kernel + (" -> neuron lists (K3n)" if lists else
                                                        " -> zero-skipping blocks (K3s)" if sparse else " -> dense (K3)")},
            "roofline": roof,
            "roofline_dense_kernel": None if dense_ms is None else {
                "kernel": "warp_gram_kernel (K3): every product evaluated, one launch outside the timed region",
                "bound": "mfma", "achieved": dense_flops / (dense_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": dense_flops / (dense_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "traffic": tjson.get(key), "launch_ms": dense_ms, "flops_per_launch": dense_flops},
            "breakdown_ms_per_step": {"gram": 1e3 * sum(k3) / args.steps,
                                      "K2_motion_kernels": 1e3 * (sum(k2) + sum(evs("motion_grad_lists"))) / args.steps},
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(res["per_step_ms"].items())},
            # HIP events between the sweeps of the timed region (this rank): how even they were
            "sweep_ms_on_stream": res["sweep_ms"],
            "fit_sanity": res["sanity"],
            "other_kernels": other_kernels(evs, P, K, T_loc, tjson, key, ops.halo_voxels(sz)),
        }
        if line["value"] is None:
            line["value_of_the_invalid_run"] = checked_value(res, args.steps)[1]
            line["invalid"] = "fit_sanity.ok is false: the sweeps did not work on a sane fit, the rate is not a measurement"
        if tjson.note():
            line["traffic_note"] = tjson.note()
        if world == 1 and not args.no_cpu_baseline and Z == 1:
            line["cpu_baseline"] = cpu_baseline(size, K, bs, args.iter_c, positions0.numpy(), frames[:bs].cpu().numpy(), res["lr"])

    # ---- secondary measurements (N = 1, default workload): never part of `value` -----------------------------------
    if world == 1 and not args.no_extras and not args.with_spatial and Z == 1 and size == 512 and K == 100:
        del res, dn, frames
        torch.cuda.empty_cache()
        extras = {}
        try:   # (after the timed region: an extra that fails must not take the headline line with it)
            # the smallest volume the reference itself can run has two slices (Demix/dNMF.py:55 divides by Z-1)
            r2 = run_sweeps(args, [size, size, 2], K, T_loc, 5, 2, 0, 1, None)
            tj = Traffic(ops.build_stamp())
            key2 = f"{size}x{size}x2x{T_loc}_K{K}"

            def roofs_depth2(r):
                P2 = 2 * P
                out = []
                k3 = r["evs"]("warp_gram_rhs_lists")
                if k3:
                    ab = 4.0 * P2 * T_loc + 4.0 * P2 * K + 4.0 * T_loc * (K * K + K)
                    out.append(dict(hbm_roof("K3n at Z = 2: lists_tilemask_kernel + warp_gram_lists_kernel<2,...> (both passes)",
                                             1e3 * sum(k3) / len(k3), ab, "frames 4PT + footprints 4PK + G, r 4T(K^2+K)",
                                             tj.get(key2 + "_lists")), valu_issue=tj.get(key2 + "_lists_valu")))
                out += other_kernels(r["evs"], P2, K, T_loc, tj, key2, ops.halo_voxels([size, size, 2]))
                return out

            extras["depth2"] = dict(short_line(r2, 5, roofs_depth2), workload=f"{size}x{size}x2x{T_loc}, K={K}: the 8-tap kernel variants")
            del r2
            torch.cuda.empty_cache()
            # the footprint update the reference leaves commented out, wired in (K7 registration + K5 + K6)
            r3 = run_sweeps(args, sz, K, T_loc, 3, 1, 0, 1, None, with_spatial=True)

            def roofs_spatial(r):
                out = []
                if r["per_step_ms"].get("image_iwarp"):   # (several launches per sweep: the flag bytes of a launch are bounded)
                    out.append(hbm_roof("image_iwarp kernels (K7: nearest warped voxel of every lattice point), all launches of a sweep",
                                        r["per_step_ms"]["image_iwarp"], 8.0 * P * T_loc, "frames read 4PT + registered frames written 4PT"))
                for name in ("spatial_accum", "spatial_accum_lists"):
                    if r["per_step_ms"].get(name):
                        out.append(hbm_roof(f"{name} (K5: A1 = Y_i C^T over the frames)", r["per_step_ms"][name],
                                            4.0 * P * T_loc + 4.0 * K * T_loc + 4.0 * P * K,
                                            "registered frames 4PT + traces 4KT read, A1 4PK written"))
                return out

            extras["with_spatial"] = dict(short_line(r3, 3, roofs_spatial), workload="the default sweep + update_footprints(live_spatial=True)")
            del r3
            torch.cuda.empty_cache()
            # the same sweep from warps a few voxels off the identity (longer tile lists, K7 windows beyond one cell)
            r3 = run_sweeps(args, sz, K, T_loc, 5, 2, 0, 1, None, beta_init="displaced")
            extras["displaced_beta"] = dict(short_line(r3, 5), workload="the default sweep started from smooth warps a few voxels "
                                            "off the identity (bench.py: displaced_beta) instead of the identity")
            del r3
            torch.cuda.empty_cache()
            # what demo.py really drives: a stock DataLoader over a host video (PCIe-inclusive, host-bound)
            dl = {}
            for name, s2, K2, T2 in (("config2_256x256x1000_K50", 256, 50, 1000), ("config3_subset_512x512x400_K100", 512, 100, 400)):
                r4 = run_sweeps(args, [s2, s2, 1], K2, T2, 2, 1, 0, 1, None, loader="dataloader")
                dl[name] = short_line(r4, 2)
                del r4
                torch.cuda.empty_cache()
            r5 = run_sweeps(args, sz, K, T_loc, 5, 2, 0, 1, None, loader="dataset")
            dl["simulated_video_dataset_512x512x4000_K100"] = dict(
                short_line(r5, 5), note="the same stock DataLoaders over the library's SimulatedVideoDataset (host video, as in "
                                        "demo.py): frames come from dataset.device_frames(), only the loaders' index batches "
                                        "are drawn")
            del r5
            torch.cuda.empty_cache()
            # footprint values below a floor left out of the neuron lists (dnmf_amd/Demix/dNMF.py: footprint_floor): the same
            # fit, the same number of sweeps, compared with the exact-support fit
            ff = {}
            ref = run_sweeps(args, sz, K, T_loc, 5, 2, 0, 1, None)
            C_ref, b_ref = ref["dn"].C.clone(), ref["dn"].fp.beta.detach().clone()
            del ref
            torch.cuda.empty_cache()
            def deviation(r):
                dC = (r["dn"].C - C_ref).abs()
                return dict(max_trace_deviation_over_max_trace=float(dC.max() / C_ref.abs().max()),
                            max_rel_trace_deviation=float((dC / C_ref.abs().clamp_min(1e-30)).max()),
                            max_abs_beta_deviation=float((r["dn"].fp.beta.detach() - b_ref).abs().max()))

            # the yardstick: the SAME exact-support fit with the Gram data from the dense kernel K3 (every product, MFMA, another
            # order of the fp32 sums: a few units in the last place per entry).  The alternating fit amplifies such a difference
            # from sweep to sweep, whatever its source.
            gram0 = args.gram
            args.gram = "dense"
            try:
                r6 = run_sweeps(args, sz, K, T_loc, 5, 2, 0, 1, None)
            finally:
                args.gram = gram0
            ff["yardstick_exact_support_other_summation_order"] = dict(deviation(r6), what="the same fit with the dense Gram kernel K3")
            del r6
            torch.cuda.empty_cache()
            for floor in (1e-20, 1e-10):
                r6 = run_sweeps(args, sz, K, T_loc, 5, 2, 0, 1, None, footprint_floor=floor)
                ff[f"{floor:g}"] = dict(short_line(r6, 5), **deviation(r6), listed_neurons_per_voxel=r6["dn"].fp.packed_lists()["boxfrac"])
                del r6
                torch.cuda.empty_cache()
            extras["footprint_floor"] = dict(ff, note="NOT the headline: the default keeps every non-zero footprint value (a Gaussian "
                                             "footprint is a non-zero fp32 number out to 30 voxels).  Deviations are against the "
                                             "exact-support fit after the same 7 sweeps: summation order, not lost terms")
            del C_ref, b_ref
            extras["position_initialiser"] = position_initialiser_extra(size)
            extras["stock_dataloader"] = dict(dl, note="torch.utils.data.DataLoader(batch 4, shuffle, num_workers=0) over a host "
                                                       "copy of the video, as demo.py:33-35; every sweep serves the video twice "
                                                       "from the host (update_motion, update_footprints)")
        except Exception as err:   # noqa: BLE001
            import traceback
            extras["error"] = f"{type(err).__name__}: {err}"
            traceback.print_exc(file=sys.stderr)
        line["extras"] = extras
    # N > 1: who took part.  One all-gather of (rank, device ordinal, PCI bus id) over the process group the sweeps ran
    # under: N distinct devices = N RCCL ranks, without NCCL_DEBUG.
    if world > 1:
        import torch.distributed as dist
        props = torch.cuda.get_device_properties(dev_index)
        mine = torch.tensor([rank, dev_index, int(getattr(props, "pci_bus_id", -1)), int(getattr(props, "pci_device_id", -1))],
                            dtype=torch.int64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine, group=group)
        if rank == 0:
            rows = [[int(v) for v in t.tolist()] for t in allr]
            line["ranks"] = {"world_size": dist.get_world_size(group), "backend": dist.get_backend(group),
                             "rank_device_pcibus_pcidev": rows,
                             "distinct_devices": len({(r[1], r[2], r[3]) for r in rows})}
    # N > 1, --exchange-extra: the sweep above shards without any collective; the path's one real exchange (the footprint
    # update's all-reduce) is exercised here, outside the timed region, under a watchdog.  A rank that fails writes its
    # error where the others' watchdogs see it; then rank 0 prints the line WITH the error and every rank leaves with exit
    # code 3 -- a process that has touched the GPU and gives up must not look like a clean run.
    if world > 1 and args.exchange_extra and not args.with_spatial and Z == 1 and K <= 128:
        import glob
        import tempfile
        tag = os.path.join(tempfile.gettempdir(), f"dnmf_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        finished = threading.Event()

        def bail(reason):
            if finished.is_set():
                return
            if rank == 0:
                line.setdefault("extras", {})["sharded_footprint_update"] = {"error": reason}
                print(json.dumps(line), flush=True)
            os._exit(3)

        def watch():
            t_end = time.time() + 240.0
            while not finished.is_set():
                errs = sorted(glob.glob(tag + "_err_*"))
                if errs:
                    time.sleep(0.2)   # let the writer finish
                    bail("; ".join(open(f).read().strip()[:300] for f in errs))
                if time.time() > t_end:
                    bail("no answer within 240 s (a collective did not return)")
                time.sleep(0.5)

        if rank == 0:                      # error notes of an earlier run under the same launcher
            for f in glob.glob(tag + "_err_*"):
                os.remove(f)
        torch.distributed.barrier()
        watcher = threading.Thread(target=watch, daemon=True)
        watcher.start()
        ex = None
        try:
            ex = sharded_footprint_update(dn, frames, sz, bs, args.iter_c, group, world)
        except Exception as err:   # the other ranks are stuck in the collective: tell their watchdogs, leave through ours
            print(f"[bench] rank {rank}: sharded_footprint_update failed: {err!r}", file=sys.stderr, flush=True)
            with open(f"{tag}_err_{rank}", "w") as f:
                f.write(f"rank {rank}: {err!r}")
            time.sleep(300.0)          # the watcher thread ends the process
        finished.set()
        if rank == 0 and ex is not None:
            line.setdefault("extras", {})["sharded_footprint_update"] = ex
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
