#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X: rasterizer forward + backward (+ the caller's
activation / Adam step that BASELINE config C3 names) on synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C1]

One STEP = one optimiser iteration over one 1920x1080 view per GPU of the C3 scene (2 M Gaussians, seed 3,
SH degree 0 -- the product setting): activations (sigmoid / exp / normalize / cat)
-> GaussianRasterizer forward (HIP, through the C ABI) -> synthetic upstream gradients dL/dcolor, dL/dacc
 injected into autograd -> rasterizer backward (HIP) -> activation backward -> Adam step with the
reference's groups / learning rates.  Activations and Adam run as the fused HIP kernels of csrc/optimizer.hip
(SURVEY.md 8(f) "next" row 1) unless --torch-optimizer selects the reference's separate Torch ops.
Inputs are resident in HBM before the timed region.  value = Mpixels/s of the whole job.

N > 1: one rank per GPU over RCCL, either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or -- when WORLD_SIZE is not set -- by
bench.py ITSELF: the parent starts N children before touching the GPU (gs_livm_amd.multiview.launch_local_ranks),
relays rank 0's JSON line and exits non-zero if any rank fails; `n_gpus` in the line is always the number of ranks
that really ran.  Rank r renders view r of the same scene (yaw offsets of SURVEY.md 8(d)).  Default --sync-mode scatter
= BASELINE config 4 as worded: the flat Gaussian buffer is broadcast from rank 0 ONCE before the loop (timed and
reported as `broadcast_ms`), then every rank runs the full step on its own view with no collective per step (the path
has no exchange step).  --sync-mode allreduce sums the per-view gradients with one all-reduce of the flat gradient
buffer (112 MB at M = 1) per step and every rank applies the identical Adam step to its replica;
--sync-mode owner = reduce to rank 0 + parameter broadcast.  With N > 1 the default run times `scatter` (the headline
`value`) AND `allreduce` back to back and reports both under "sync_modes" on the one JSON line.  The rasterizer itself
never communicates.  scaling = "weak".

Extra objects on the JSON line: "roofline" (dominant kernel: its bound, the live hipEvent-measured launch time over
the timed region, algorithmic bytes vs 8 TB/s under "hbm", VALU issue rate where the kernel is VALU-bound),
"cpu_baseline" (the CPU oracle on the same workload, host cores, rank 0 at N = 1 only), "kernels" (per-kernel ms per
step), "whole_path", "workload_stats" (measured P_vis, R, list lengths) and "work_units".
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import gs_livm_amd as G  # noqa: E402
from gs_livm_amd import multiview as MV  # noqa: E402
from gs_livm_amd import synthetic as S  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(P, P_vis, R, R_bwd, W, H, M, tiles):
    """Algorithmic HBM bytes per LAUNCH of every kernel (DESIGN.md section 5 states each figure).
    R = instances, R_bwd = sum over tiles of tile_last (list entries the backward has to walk)."""
    kb = 2 if tiles <= 65536 else 4
    return {
        # in: 56 B @M=1; out: radii 4, gpack (tile count + rect) 8, touched 1, clamped 1, depth-sort key 4 per Gaussian
        # and the 48-B splat record per visible one (cov3D, a separate depth array and the id array are no longer written)
        "k_preprocess": P * (44 + 12 * M) + P * 18 + P_vis * 48,
        "k_point_offsets": P * 8,                                 # debug forwards only
        "k_sort_hist[depth]": P * 4, "k_sort_scatter[depth]": P * 16,
        # order + gpack gather in; 16-byte depth-ordered descriptor out; slotinfo for the Gaussians with instances
        "k_scan_offsets": P * 12 + P * 16 + P_vis * 8,
        # tile ids are 16-bit when the image has <= 65536 tiles (kb bytes per key), Gaussian ids 32-bit
        # 16-bit tile ids: the emitter only counts digits and clears flags (the pairs are generated inside the first
        # sort pass, which therefore reads descriptors instead of pairs): per-launch AVERAGE of the two scatter passes
        "k_emit": (R * 1 + P_vis * 16) if kb == 2 else (R * (kb + 4) + P_vis * 16),
        "k_sort_hist": R * kb,
        # (first pass: descriptors in, pairs out; second pass: pairs in, ids out -- the sorted tile ids are not written)
        "k_sort_scatter": (R * ((kb + 4) * 2 + 4) + P_vis * 16) // 2 if kb == 2 else R * 2 * (kb + 4),
        # 16-bit keys: ranges come from the last pass's per-tile counts (scan of T counters), no key read
        "k_tile_ranges": tiles * 16 if kb == 2 else R * kb + tiles * 8,
        "k_sort_scan_chunks": (R // 4096 + 1) * 2048, "k_sort_scan_top": 0,
        # SURVEY.md 8(d)'s 44 B per gathered instance x the entries the kernel has to walk (per tile: up to its last
        # contributing entry, the same count the backward walks -- NOT the full lists: saturated pixels stop early and
        # charging 44 R would credit the kernel with bytes it never needs)
        "k_blend_forward": R_bwd * 44 + W * H * 28,
        "k_blend_backward": W * H * 24 + R_bwd * (40 + 36),       # SURVEY.md 8(d) per-instance figures x walked entries
        "k_compact_touched": P * 1 + P_vis * 0,
        "k_gather_records": R_bwd * 48,
        # what the kernel moves (every access is unconditional, culled Gaussians included): reads radii 4, touched 1,
        # clamp flags 1, the four gathered sums 12 + 8 + 12 + 4, mean 12, scale 12, rotation 16, SH rows 12 M; writes the
        # nine gradient arrays in full, 108 + 12 M (zeros for culled Gaussians: callers need no memset).  PMC at C3:
        # 447 MB against 436 MB by this formula (the earlier figure also charged SURVEY's zero fill and a cov3D read)
        "k_gaussian_backward": P * (82 + 12 * M) + P * (108 + 12 * M),
        "k_compact_near": P * 4,                                  # (the near candidates written are a few per cent)
        # "next" row kernels: activations read/write the 14 + 3(M-1)... floats per Gaussian once each way
        "k_activate": P * 4 * (11 + 3 * M) * 2 - P * 24,          # xyz is not touched
        "k_activate_backward": P * 4 * (8 + 3 * M) * 2 + P * 4 * 8,
        "k_adam": P * 4 * (11 + 3 * M) * 7,                       # p, g, m, v read; p, m, v (+ zeroed g) written
        # one-kernel tail: p, m, v, g (activated space) read; p, m, v + the activated values written
        "k_model_step": P * 4 * (11 + 3 * M) * 7 + P * 4 * (8 + 3 * M),
        "k_loss_forward": W * H * 3 * 4 * 5, "k_loss_backward": W * H * 3 * 4 * 6, "k_loss_finalize": 0,
    }


VALU_PEAK_GINST = 1024 * 2.4 / 2.0  # G wave-instructions/s: 1024 SIMD-32s x 2.4 GHz, one wave64 VALU op per 2 cycles


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=sorted(S.CONFIGS))
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adam", action="store_true", help="time rasterizer fwd+bwd only (diagnostic)")
    ap.add_argument("--no-fused-tail", action="store_true",
                    help="activations backward, Adam and next activations as three fused kernels instead of one "
                         "(gsr_model_step); the one-kernel tail is used whenever gradients are not exchanged")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="activations / Adam as separate Torch ops (what the reference does) instead of the fused kernels")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE C2 style: colour+depth+silhouette forward only")
    ap.add_argument("--host", default="cpp", choices=("cpp", "python"),
                    help="operator surface used for the rasterizer: the C++/LibTorch binding GS-LIVM would link "
                         "(csrc/torch_binding.cpp) or the Python mirror over ctypes")
    ap.add_argument("--loss", default="seeded", choices=("seeded", "photometric"),
                    help="seeded: inject the fixed upstream gradients of SURVEY.md 8(d); photometric: the reference's "
                         "0.8*L1 + 0.2*(1-SSIM) against a fixed random target (fused HIP loss kernels)")
    ap.add_argument("--sync-mode", default="scatter", choices=("scatter", "allreduce", "owner"),
                    help="N > 1: 'scatter' (BASELINE C4 as worded: independent views, RCCL only scatters the shared "
                         "Gaussian buffer -- one broadcast from rank 0 before the loop, no collective per step); "
                         "'allreduce' = joint optimisation of all views: sum the per-view gradients on every rank and "
                         "step replicated Adam (one 56 B/Gaussian collective per step); 'owner' = reduce to rank 0, "
                         "Adam there, broadcast the parameters (two collectives per step)")
    ap.add_argument("--single-sync-mode", action="store_true",
                    help="N > 1: time only --sync-mode (default: 'scatter' is followed by an 'allreduce' run, both reported)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo = rehearsal of the multi-rank control flow (collectives staged through the host)")
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="the reference's loop shape (lioOptimization.cpp:1691-1737, 1822-1832): K forwards of DIFFERENT "
                         "views per optimiser iteration (the C4 yaw cameras in turn, advancing from step to step), their "
                         "losses summed, ONE backward, one Adam step.  value counts every rendered view's pixels; "
                         "ms_per_step is the whole iteration")
    ap.add_argument("--view-threads", type=int, default=1,
                    help="render the K views of an iteration from T host threads, each on a stream of its own (the "
                         "reference renders from several threads, SURVEY.md 8b): one view's latency-bound binning chain "
                         "then runs beside another view's blend kernels; the backwards follow on the same streams")
    ap.add_argument("--rotate-views", action="store_true",
                    help="K = 1: the camera still changes on every step (default: one fixed view, BASELINE C3)")
    ap.add_argument("--grow-every", type=int, default=0,
                    help="every n-th step GrowableGaussians.add_new_pointcloud adds --grow-points Gaussians in place "
                         "(gaussian.cu:241-313; P grows every few iterations in the reference's loop); 1 GPU only")
    ap.add_argument("--grow-points", type=int, default=20000)
    ap.add_argument("--opacity-scale", type=float, default=1.0,
                    help="multiplies the scene's opacities (0.1: a scene whose tiles never saturate inside the near budget "
                         "-- the one-chain floor of the forward)")
    ap.add_argument("--near-entries", type=int, default=0,
                    help="diagnostic: fix the near budget at this many list entries per tile (the adaptive rule is off)")
    ap.add_argument("--per-step", action="store_true",
                    help="after the timed region: an extra pass of the same steps, synchronised one by one, whose ms and "
                         "speculation counters per step go into the line under 'per_step' (diagnostic, never `value`)")
    ap.add_argument("--reference-rects", action="store_true",
                    help="bin with the reference's full 3-sigma tile squares (gsr_set_reference_rects(1)) instead of the "
                         "footprint-culled default: the reference's own instance count through the whole path")
    return ap.parse_args()


class _JsonLinesToStdout:
    """rank 0's stdout as the parent relays it: JSON lines to stdout, anything else (library chatter) to stderr."""

    def write(self, text):
        for line in text.splitlines():
            print(line, file=sys.stdout if line.startswith("{") else sys.stderr)

    def flush(self):
        sys.stdout.flush()


def self_launch(args):
    """--gpus N without an external launcher: N children, one per GPU.  Nothing here may initialise the GPU
    (torch.cuda.device_count() does not)."""
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and ndev < args.gpus:
        print("bench.py --gpus %d: only %d GPU(s) visible and RCCL needs one per rank (a rehearsal of the multi-rank "
              "control flow on fewer GPUs: --dist-backend gloo)" % (args.gpus, ndev), file=sys.stderr)
        return 2
    print("[bench] starting %d ranks (one per GPU, backend %s) on 127.0.0.1" % (args.gpus, args.dist_backend),
          file=sys.stderr)
    return MV.launch_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                 out=_JsonLinesToStdout())


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):  # never bench a different rank count than the one asked for
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    local_rank %= max(1, torch.cuda.device_count())  # a gloo rehearsal may put several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if rank == 0:
            print("[bench] process group up: backend %s, %d ranks" % (dist.get_backend(), dist.get_world_size()),
                  file=sys.stderr)
    n_gpus = world
    G.set_reference_rects(args.reference_rects)
    if args.near_entries > 0:  # (main thread only: a measurement knob for single-thread runs)
        G.set_near_far_hints(near_entries_per_tile=args.near_entries)

    P, W, H, seed = S.CONFIGS[args.workload]
    D = args.sh_degree
    M = (D + 1) ** 2
    g = S.make_gaussians(P, seed, sh_degree=D, aspect=W / H)
    if args.opacity_scale != 1.0:
        g["opacities"] = (g["opacities"] * np.float32(args.opacity_scale)).astype(np.float32)
    K = max(1, args.views_per_step)
    rotate = K > 1 or args.rotate_views
    if args.grow_every and (n_gpus > 1 or args.torch_optimizer or args.no_adam or args.forward_only):
        raise SystemExit("bench.py: --grow-every needs the fused optimiser on one GPU")
    yaw = S.C4_YAWS_DEG[rank % len(S.C4_YAWS_DEG)] if n_gpus > 1 else 0.0
    cam = S.make_camera(W, H, yaw_deg=yaw)
    # the cameras a rank cycles through when the view changes from forward to forward (--views-per-step / --rotate-views)
    cams = [S.make_camera(W, H, yaw_deg=y) for y in S.C4_YAWS_DEG] if rotate else [cam]
    pre = dict(means3D=g["means3D"], scales=np.log(g["scales"]), rotations=g["rotations"],
               opacities=np.log(g["opacities"] / (1.0 - g["opacities"])), shs=g["shs"])
    pre = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in pre.items()}

    bg = torch.ones(3, device=dev)
    settings = G.GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], bg, 1.0,
                                               torch.from_numpy(cam["viewmatrix"]).to(dev),
                                               torch.from_numpy(cam["projmatrix"]).to(dev), D,
                                               torch.from_numpy(cam["campos"]).to(dev), False)
    def make_raster(cm):
        vm, pm, cc = (torch.from_numpy(cm[k]).to(dev) for k in ("viewmatrix", "projmatrix", "campos"))
        if args.host == "cpp":
            T = G.torch_ops()
            r_cpp = T.GaussianRasterizer(T.GaussianRasterizationSettings(
                H, W, cm["tanfovx"], cm["tanfovy"], bg, 1.0, vm, pm, D, cc, False))
            return lambda xyz, m2d, op, shs, scales, rotations: r_cpp.forward(xyz, m2d, op, shs=shs, scales=scales,
                                                                              rotations=rotations)
        return G.GaussianRasterizer(G.GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], bg, 1.0, vm, pm,
                                                                    D, cc, False))

    rasters = [make_raster(cm) for cm in cams]
    raster = rasters[0]
    dcol, dacc = S.make_upstream_grads(W, H, seed + rank)
    wc, wa = torch.from_numpy(dcol).to(dev), torch.from_numpy(dacc).to(dev)
    # gradient sinks, one per view of an iteration, as render_utils.cuh:39-40 (re-made when the model grows)
    sinks = {"P": P, "t": [torch.zeros((P, 3), device=dev, requires_grad=True) for _ in range(K)]}
    means2D = sinks["t"][0]
    target = torch.rand((3, H, W), generator=torch.Generator().manual_seed(seed)).to(dev)  # --loss photometric
    window = G.reference_window_1d()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def run_mode(sync_mode, survey):
        """W warm-up steps, (survey: an untimed pass with hipEvents around every kernel,) then EXACTLY K timed steps
        between barrier + synchronize pairs; returns the max-over-ranks time and what the stats need."""
        # ---- leaf parameters in one flat buffer (the "Gaussian buffer" of BASELINE C4), pre-activation ----
        params = MV.GaussianBuffer(P, M, dev)
        grads = MV.GaussianBuffer(P, M, dev)
        if rank == 0:
            params.load(pre)
        barrier()
        t0 = time.perf_counter()
        MV.broadcast_gaussians(params, src=0)  # C4's one exchange: the shared Gaussian buffer to every GPU
        barrier()
        broadcast_ms = max_over_ranks((time.perf_counter() - t0) * 1e3) if dist is not None else None
        v = params.views
        grow = args.grow_every > 0
        if grow:
            # the model as capacity buffers that grow in place: room for every growth of this run
            n_grow = (args.warmup + args.steps * (3 if args.per_step else 2)) // args.grow_every + 2
            model = G.GrowableGaussians(P + n_grow * args.grow_points, M, dev)
            model.P = P
            model._bind()
            with torch.no_grad():
                for name, key in (("_xyz", "means3D"), ("_features_dc", "features_dc"), ("_features_rest", "features_rest"),
                                  ("_scaling", "scales"), ("_rotation", "rotations"), ("_opacity", "opacities")):
                    getattr(model, name).copy_(v[key])
            grow_rng = torch.Generator(device=dev).manual_seed(seed + 77)
        else:
            model = G.GaussianParameters(v["means3D"], v["features_dc"], v["features_rest"], v["scales"],
                                         v["rotations"], v["opacities"])
        leaves = dict(means3D=model._xyz, features_dc=model._features_dc, features_rest=model._features_rest,
                      scales=model._scaling, rotations=model._rotation, opacities=model._opacity)
        joint = n_gpus > 1 and sync_mode != "scatter"  # views optimised jointly: gradients are exchanged
        flat_grads = joint   # one flat gradient buffer = one collective; otherwise autograd hands over its
        if flat_grads:           # gradient tensors as they are (no accumulate kernels, nothing to zero)
            for k, p in leaves.items():
                p.grad = grads.views[k]
        # groups / learning rates of GaussianModel::Training_setup (src/gs/gaussian.cu:396-428) with the values of
        # config/basic_common.yaml:54-62, eps 1e-15
        groups = [gr for gr in model.param_groups() if gr["params"][0].numel()]
        if args.torch_optimizer:
            opt = torch.optim.Adam(groups, eps=1e-15, fused=True)
        elif grow:
            opt = G.GrowableAdam(model, eps=1e-15)
        else:
            opt = G.FusedAdam(groups, eps=1e-15)
        counter = {"step": 0, "grown": 0}
        # --view-threads T: persistent rendering threads, each with a stream of its own and (inside the library) its own
        # per-thread state; view j of an iteration always goes to thread j % T
        T_views = max(1, min(args.view_threads, K))
        workers = MV.ViewThreads(T_views, dev) if T_views > 1 else None

        def grow_model():
            """addNewPointcloud (gaussian.cu:241-313): --grow-points new Gaussians from the scene's own distribution."""
            # (generated on the device: a burst of multi-threaded CPU work here can use up the container's CPU quota and
            # get every host thread descheduled for the rest of the scheduler period -- see DESIGN.md, "mailbox")
            n = args.grow_points
            r = lambda *shape: torch.rand(shape, generator=grow_rng, device=dev)  # noqa: E731
            z = 1.0 + 39.0 * r(n)
            t = math.tan(math.radians(30.0))
            xyz = torch.stack([(2 * r(n) - 1) * 1.1 * z * t, (2 * r(n) - 1) * 1.1 * z * t * H / W, z], 1)
            covs = torch.diag_embed((0.004 + 0.056 * r(n, 3)) ** 2)
            rgbs = 255.0 * r(n, 3)
            model.add_new_pointcloud(xyz, covs, rgbs, 1.0)
            sinks["P"] = model.P
            sinks["t"] = [torch.zeros((model.P, 3), device=dev, requires_grad=True) for _ in range(K)]
            counter["grown"] += n

        def activated():
            if args.torch_optimizer:  # the reference's getters as separate Torch ops (gaussian.cuh:40-54)
                return (model._xyz, torch.sigmoid(model._opacity), torch.exp(model._scaling),
                        torch.nn.functional.normalize(model._rotation, dim=1),
                        torch.cat([model._features_dc, model._features_rest], 1))
            return model.activated()

        owner_mode = joint and sync_mode == "owner"
        tail = not (joint or args.torch_optimizer or args.no_adam or args.no_fused_tail or args.forward_only)
        model.fused_tail = tail

        def step():
            it = counter["step"]
            counter["step"] += 1
            if grow and it and it % args.grow_every == 0:
                grow_model()
            if owner_mode:
                MV.broadcast_gaussians(params, src=0)
            xyz, op, sc, rot, shs = activated()
            # this iteration's views: K different cameras, advancing from step to step (one fixed camera otherwise)
            views = [rasters[(it * K + j) % len(rasters)] for j in range(K)] if rotate else [raster] * K
            if args.forward_only:
                with torch.no_grad():
                    for j, rv in enumerate(views):
                        rv(xyz, sinks["t"][j], op, shs=shs, scales=sc, rotations=rot)
                return
            if workers is None:
                outs = [rv(xyz, sinks["t"][j], op, shs=shs, scales=sc, rotations=rot) for j, rv in enumerate(views)]
            else:
                outs = workers.render([lambda rv=rv, j=j: rv(xyz, sinks["t"][j], op, shs=shs, scales=sc, rotations=rot)
                                       for j, rv in enumerate(views)])
            for m2d in sinks["t"]:
                m2d.grad = None
            if args.torch_optimizer and flat_grads:
                grads.flat.zero_()
            # K forwards, the losses summed, ONE backward (lioOptimization.cpp:1705-1710, 1822-1825)
            if args.loss == "photometric":  # 0.8 L1 + 0.2 (1 - SSIM), lambda_dssim = 0.2
                loss = G.photometric_loss(outs[0][0], target, 0.2, window)
                for o in outs[1:]:
                    loss = loss + G.photometric_loss(o[0], target, 0.2, window)
                loss.backward()
            else:  # upstream gradients injected directly (SURVEY.md 8(d) backward seeds): dL/dcolor = wc, dL/dacc = wa
                torch.autograd.backward([t for o in outs for t in (o[0], o[3])], [wc, wa] * len(outs))
            if joint:
                MV.reduce_gradients(grads, dst=0, all_ranks=not owner_mode)
            if tail:
                opt.step_model(model)
            elif not args.no_adam and (rank == 0 or not owner_mode):
                if args.torch_optimizer:
                    opt.step()
                else:
                    opt.step(zero_grads=flat_grads)  # clears the flat buffer it consumed; replicas stay identical
            elif flat_grads:
                grads.flat.zero_()
            if not flat_grads:
                for p in leaves.values():
                    p.grad = None

        for _ in range(args.warmup):
            step()
        barrier()
        prof, dom_name = None, None
        if survey:
            # Untimed survey pass: HIP events around EVERY kernel (the `kernels` table).  An event pair costs ~5 us of
            # GPU idle per launch, so the timed region below carries events for the dominant kernel only -- the one
            # the roofline object is about.
            G.profile_enable(True)
            for _ in range(args.steps):
                step()
            barrier()
            G.profile_enable(False)
            prof = G.profile_read()
            dom_name = max((k for k, (ms, c) in prof.items() if c), key=lambda k: prof[k][0])
            G.profile_enable(True, only=[dom_name])
        spec0 = G.speculation_stats()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        spec1 = G.speculation_stats()
        # what the history-driven fast paths did over the TIMED region: a mispredicting run cannot hide
        speculation = {k: spec1[k] - spec0[k] for k in spec1 if k != "near_budget_scale_q8"}
        speculation["near_budget_scale_q8"] = spec1["near_budget_scale_q8"]
        speculation["forwards"] = args.steps * K
        prof_timed = None
        if survey:
            G.profile_enable(False)
            prof_timed = G.profile_read()
        per_step = None
        if args.per_step:  # diagnostic pass, synchronised step by step (never part of `value`)
            per_step = []
            for _ in range(args.steps):
                a = G.speculation_stats()
                torch.cuda.synchronize()
                ma = torch.cuda.memory_stats(dev)
                ts = time.perf_counter()
                step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - ts) * 1e3
                b = G.speculation_stats()
                mb = torch.cuda.memory_stats(dev)
                per_step.append(dict(ms=round(ms, 4), overflows=b["overflows"] - a["overflows"],
                                     # trips of the caching allocator to the driver during this step (a multi-GB blob: ~10 ms)
                                     device_allocs=mb.get("num_device_alloc", 0) - ma.get("num_device_alloc", 0),
                                     device_frees=mb.get("num_device_free", 0) - ma.get("num_device_free", 0),
                                     slow_path=b["mailbox_slow_path_hits"] - a["mailbox_slow_path_hits"],
                                     far_skips=b["far_skips"] - a["far_skips"],
                                     far_skip_misses=b["far_skip_misses"] - a["far_skip_misses"],
                                     near_budget_scale_q8=b["near_budget_scale_q8"], P=int(model._xyz.shape[0])))
        if workers is not None:
            workers.close()
        return dict(elapsed=elapsed, prof=prof, prof_timed=prof_timed, dom_name=dom_name, activated=activated,
                    tail=tail, broadcast_ms=broadcast_ms, speculation=speculation, per_step=per_step,
                    grown=counter["grown"], P_end=int(model._xyz.shape[0]), collective_bytes_per_step=(
                        0 if not joint else params.nbytes() * (2 if owner_mode else 1)))

    main_run = run_mode(args.sync_mode, survey=True)
    elapsed, prof, prof_timed, dom_name = (main_run[k] for k in ("elapsed", "prof", "prof_timed", "dom_name"))
    activated, tail = main_run["activated"], main_run["tail"]
    P0, P = P, main_run["P_end"]  # (--grow-every: the statistics below describe the model as the run left it)

    # ---- workload statistics from one direct forward with the current parameters ----
    with torch.no_grad():
        xyz, op, sc, rot, shs = [t.contiguous() for t in activated()]
        e = torch.empty(0, device=dev)
        fw = G.rasterize_forward(bg, xyz, e, op, sc, rot, 1.0, e, settings.viewmatrix, settings.projmatrix,
                                 settings.tanfovx, settings.tanfovy, H, W, shs, D, settings.camera_center)
        R, radii = fw[0], fw[4]
        nf_split, nf_near, nf_far = G.last_near_far()  # was this frame binned near / far, and the two chains' instances
        P_vis = int((radii > 0).sum())
        v = G.state_views(fw[5], fw[6], fw[7], P, R, W, H)
        ln = (v["ranges"][:, 1] - v["ranges"][:, 0]).float()
        walk = v["quad_last"].long().max(1).values  # list entries the backward walks, per tile
        R_bwd = int(walk.sum())
        # instance count of the REFERENCE's binning (getRect on the 3-sigma radius, auxiliary.h:46-57) for the same
        # frame: the footprint-box culling emits fewer
        vis = radii > 0
        px, py, rr = v["splats"][vis, 0], v["splats"][vis, 1], radii[vis].float()
        gx, gy = (W + 15) // 16, (H + 15) // 16
        x0 = ((px - rr) / 16).trunc().clamp(0, gx); x1 = ((px + rr + 15) / 16).trunc().clamp(0, gx)
        y0 = ((py - rr) / 16).trunc().clamp(0, gy); y1 = ((py + rr + 15) / 16).trunc().clamp(0, gy)
        R_ref = int(((x1 - x0) * (y1 - y0)).double().sum())
        stats = dict(P=P, P_vis=P_vis, R=R, R_reference_binning=R_ref, R_walked_by_backward=R_bwd, tiles=int(ln.numel()),
                     mean_tile_list=float(ln.mean()), max_tile_list=int(ln.max()),
                     mean_contrib_per_pixel=float(v["n_contrib"].float().mean()),
                     backward_walk_per_tile=dict(mean=float(walk.float().mean()), max=int(walk.max()),
                                                 p50=int(walk.float().quantile(0.5)), p90=int(walk.float().quantile(0.9)),
                                                 p99=int(walk.float().quantile(0.99))),
                     binning="reference rectangles" if args.reference_rects else "footprint-culled (default)",
                     near_far=dict(split=bool(nf_split), near_instances=nf_near, far_instances=nf_far))
        del fw, v
    tiles = stats["tiles"]
    alg = algorithmic_bytes(P, P_vis, R, R_bwd, W, H, M, tiles)
    kernels = {}
    for name, (ms, cnt) in prof.items():
        if cnt:
            kernels[name] = dict(ms_per_step=ms / args.steps, launches_per_step=cnt / args.steps,
                                 avg_launch_ms=ms / cnt,
                                 alg_GBps=(alg.get(name, 0) / (ms / cnt * 1e-3) / 1e9) if ms > 0 else None)
    dom = dom_name if alg.get(dom_name) else max((k for k in kernels if alg.get(k)),
                                                 key=lambda k: kernels[k]["ms_per_step"])
    if prof_timed.get(dom, (0, 0))[1]:  # the dominant kernel's duration over the TIMED region
        ms, cnt = prof_timed[dom]
        kernels[dom].update(ms_per_step=ms / args.steps, avg_launch_ms=ms / cnt,
                            alg_GBps=alg[dom] / (ms / cnt * 1e-3) / 1e9)
    launch_ms = kernels[dom]["avg_launch_ms"]
    hbm_achieved = alg[dom] / (launch_ms * 1e-3) / 1e9
    # Committed counter summaries (PMC collection cannot run inside a timed bench): used ONLY when they were taken
    # on this workload with these flags (their "_meta"), otherwise the fields stay null.
    this_meta = dict(workload=args.workload, sh_degree=D, loss=args.loss, forward_only=bool(args.forward_only),
                     reference_rects=bool(args.reference_rects), n_gpus=n_gpus)
    plain_run = K == 1 and not rotate and not args.grow_every and args.opacity_scale == 1.0 and not args.near_entries

    def committed(fname):
        try:
            with open(os.path.join(ROOT, "profiles", fname)) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            return None
        meta = d.get("_meta") or {}
        return d if plain_run and all(meta.get(k) == val for k, val in this_meta.items()) else None

    def source_of(fname, d):
        """Where a counter figure of this line comes from: PMC collection cannot run inside the timed bench, so these
        are read from the committed summary of a counter pass over the same workload -- said so in the line."""
        if not d:
            return None
        return "profiles/%s (%s; counter passes run with GSR_ASYNC_FAR=0: host-decided far chain)" % (
            fname, (d.get("_meta") or {}).get("tag", "committed counter pass"))

    pmc = committed("pmc_traffic_latest.json")
    # device-side names behind one profiler id (the blend backward is one of two kernels, chosen by the tile count)
    dev_names = {"k_blend_backward": ("k_blend_backward_tile", "k_blend_backward")}.get(dom, (dom,))
    traffic = next(((pmc or {})[k].get("hbm_bytes_per_launch") for k in dev_names if k in (pmc or {})), None)
    sq = committed("sq_counters_latest.json")
    ent = next((val for name in dev_names for k, val in (sq or {}).items()
                if k.split("<")[0] == name and isinstance(val, dict)), None)
    hbm = dict(achieved=round(hbm_achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_achieved / HBM_PEAK_GBS, 4),
               algorithmic_bytes_per_launch=int(alg[dom]))
    valu_bound = dom in ("k_blend_backward", "k_blend_forward")  # bound by VALU issue, not bytes
    if valu_bound and ent and ent.get("SQ_INSTS_VALU"):
        # VALU issue roofline: wave-instructions per launch (SQ_INSTS_VALU of the committed counter pass on this same
        # workload) over the LIVE launch time, against one wave64 VALU op per 2 cycles per SIMD-32
        ginst = ent["SQ_INSTS_VALU"] / (launch_ms * 1e-3) / 1e9
        roofline = dict(kernel=dom, bound="valu", achieved=round(ginst, 1), peak=VALU_PEAK_GINST,
                        unit="G wave-instr/s", frac=round(ginst / VALU_PEAK_GINST, 4), traffic=traffic,
                        traffic_source=source_of("pmc_traffic_latest.json", pmc) if traffic else None,
                        avg_launch_ms=round(launch_ms, 4), wave_instructions_per_launch=ent["SQ_INSTS_VALU"],
                        wave_instructions_source=source_of("sq_counters_latest.json", sq),
                        # the same against the rate MEASURED for plain f32 ops at 4-8 waves per SIMD
                        # (tools/microbench/valu_rate.hip: v_fma / v_mul 3.0-3.3 cycles, v_add_f32_dpp 4.4-4.7, v_exp 8.3)
                        frac_of_measured_3cycle_rate=round(ginst / (1024 * 2.4 / 3.0), 4), hbm=hbm)
    else:
        roofline = dict(kernel=dom, bound="hbm", achieved=hbm["achieved"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=hbm["frac"], traffic=traffic,
                        traffic_source=source_of("pmc_traffic_latest.json", pmc) if traffic else None,
                        avg_launch_ms=round(launch_ms, 4),
                        algorithmic_bytes_per_launch=int(alg[dom]),
                        note=("VALU-issue-bound kernel (DESIGN.md section 4); no counter summary committed for this "
                              "workload, so only its HBM figure is given") if valu_bound else None)
    # whole path: sum over kernels of (algorithmic bytes per launch x launches per step)
    b_path = sum(alg.get(k, 0) * d["launches_per_step"] for k, d in kernels.items())
    raster_ms = sum(k["ms_per_step"] for k in kernels.values())

    ms_per_step = elapsed / args.steps * 1e3
    mpix = n_gpus * K * W * H * args.steps / elapsed / 1e6  # every rendered (and differentiated) view counts
    out = {
        "metric": "rasterizer %s Mpixels/s @%dx%d, %d Gaussians (ms_per_step = ms/frame)" %
                  ("fwd" if args.forward_only else "fwd+bwd", W, H, P),
        "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d Gaussians, %dx%d, SH degree %d, %d view%s per GPU per step, %s" %
                               (args.workload, P0, W, H, D, K, "" if K == 1 else "s",
                                "forward only" if args.forward_only else
                                ("fwd+bwd + Adam step" if K == 1 else "%d forwards, one backward, one Adam step" % K)),
                   "views_per_step": n_gpus * K, "views_per_gpu_per_step": K,
                   "view_threads": max(1, min(args.view_threads, K)),
                   "camera": "the C4 yaw cameras in turn, a different one every forward" if rotate else "one fixed view",
                   "grow": (dict(every_steps=args.grow_every, points=args.grow_points, added=main_run["grown"],
                                 P_at_end=P) if args.grow_every else None),
                   "opacity_scale": args.opacity_scale, "near_entries": args.near_entries,
                   "parallelism": "view-parallel x%d" % n_gpus,
                   "sync_mode": args.sync_mode if n_gpus > 1 else None,
                   "adam_in_step": not (args.no_adam or args.forward_only),
                   "optimizer": "torch ops" if args.torch_optimizer else
                                ("one-kernel tail: activation chain rule + Adam + next activations (HIP)" if tail else
                                 "fused activations + fused Adam (HIP)"),
                   "loss": args.loss, "host": args.host},
        "fps": round(1e3 / ms_per_step * n_gpus * K, 2),
        "ms_per_view": round(ms_per_step / K, 4),
        # what the history-driven fast paths did over the timed region (speculative forwards: overflows = frames binned
        # twice; far_skips / far_skip_misses = split frames whose far chain stayed closed / had to run; near budget
        # scale in 1/256 of the configured entries per tile)
        "speculation": main_run["speculation"],
        "per_step": main_run["per_step"],
        "roofline": roofline,
        "whole_path": {"kernel_ms_per_step": round(raster_ms, 4), "algorithmic_GB_per_step": round(b_path / 1e9, 3),
                       "alg_GBps_over_kernel_time": round(b_path / (raster_ms * 1e-3) / 1e9, 1) if raster_ms else None},
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in d.items()}
                    for k, d in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
        "workload_stats": stats,
        # forwards whose instance count reached the host through the fallback stream query (0 in a healthy run)
        "mailbox_slow_path_hits": int(G.lib().gsr_mailbox_slow_path_hits()),
        # SURVEY.md 8(d) work units: instances through the binning per second of step time, and (pixel, contributor)
        # pairs per second of each blend kernel's own time (contributors = n_contrib summed over the image)
        "work_units": {
            "Minstances_per_s": round(stats["R"] / (ms_per_step * 1e-3) / 1e6, 1),
            "Gpairs_per_s_blend_forward": (round(stats["mean_contrib_per_pixel"] * W * H /
                                                 (kernels["k_blend_forward"]["avg_launch_ms"] * 1e-3) / 1e9, 2)
                                           if "k_blend_forward" in kernels else None),
            "Gpairs_per_s_blend_backward": (round(stats["mean_contrib_per_pixel"] * W * H /
                                                  (kernels["k_blend_backward"]["avg_launch_ms"] * 1e-3) / 1e9, 2)
                                            if "k_blend_backward" in kernels else None),
        },
    }
    if n_gpus > 1:
        def mode_line(r):
            ms = r["elapsed"] / args.steps * 1e3
            return dict(value=round(n_gpus * W * H * args.steps / r["elapsed"] / 1e6, 2), unit="Mpixels/s",
                        ms_per_step=round(ms, 4), collective_bytes_per_step=r["collective_bytes_per_step"],
                        broadcast_of_the_gaussian_buffer_ms=round(r["broadcast_ms"], 3),
                        gaussian_buffer_bytes=P * MV.floats_per_gaussian(M) * 4)
        out["sync_modes"] = {args.sync_mode: mode_line(main_run)}
        if args.sync_mode == "scatter" and not (args.single_sync_mode or args.forward_only or args.no_adam):
            del main_run, activated
            torch.cuda.empty_cache()
            out["sync_modes"]["allreduce"] = mode_line(run_mode("allreduce", survey=False))
        out["dist_backend"] = args.dist_backend

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and not args.forward_only:
        out["cpu_baseline"] = cpu_baseline(g, cam, dcol, dacc, W, H, D)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(g, cam, dcol, dacc, W, H, D):
    """The CPU oracle (oracle/gsr_oracle.c, a restatement -- kind 'port') on the same workload: one full frame
    forward + backward on all host cores available to this process."""
    from oracle import oracle as O
    sc = dict(g, **cam, bg=np.ones(3, np.float32), scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None)
    # the box's CPU share for one GPU is 16 cores: never spawn more workers than that
    threads = min(O.max_threads(), len(os.sched_getaffinity(0)), 16)
    O.set_threads(threads)
    t0 = time.perf_counter()
    fr = O.forward(sc)
    t1 = time.perf_counter()
    O.backward(fr, sc, dcol, dacc)
    t2 = time.perf_counter()
    fr.close()
    sec = t2 - t0
    return {"value": round(W * H / sec / 1e6, 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": "1 frame forward+backward of the same workload (%dx%d, %d Gaussians, R=%d); fwd %.2f s, bwd %.2f s"
                      % (W, H, g["means3D"].shape[0], fr.R, t1 - t0, t2 - t1),
            "ms_per_frame": round(sec * 1e3, 1)}


if __name__ == "__main__":
    main()
