#!/usr/bin/env python3
"""clips/s of the AltFreezing i3d_ori forward on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16] [--dtype bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

A step = one `model.forward(clip_batch)` of the drop-in Classifier on a (B,3,32,224,224) fp32 normalised
batch already resident in HBM (the callers' channels-last strided tensor), B = 16 clips per GPU
(BASELINE config[1]); with N > 1 every rank runs its own 16 clips and the step ends with the RCCL
all-gather of the (N*16, 1) logits (config[2], weak scaling).  Weights are the seeded synthetic checkpoint
W(0) in the reference's state_dict layout; clips are seeded synthetic uint8 face crops.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}    # dense MFMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def cpu_baseline(sd, u8, threads_all, model="i3d"):
    """The oracle (PyTorch-CPU restatement of the reference forward, pinned by tests/golden) timed on the
    host cores: a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import i3d_oracle as oracle
    x = oracle.normalize(u8)
    fwd = {"i3d": oracle.forward, "ftcn_tt": oracle.ftcn_forward,
           "slowfast": (lambda s_, x_: oracle.slowfast_forward(s_, x_[:, :, ::8], x_))}[model]
    torch.set_num_threads(threads_all)
    with torch.no_grad():
        fwd(sd, x[:1])                                              # warm-up (first call pages oneDNN in)
        t0 = time.perf_counter()
        ref = fwd(sd, x)
        t_all = time.perf_counter() - t0
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        fwd(sd, x[:1])
        t_one = time.perf_counter() - t0
    torch.set_num_threads(threads_all)
    return ref, {"value": round(x.shape[0] / t_all, 4), "unit": "clips/s", "cores": threads_all, "kind": "port",
                 "sample": "%d clips (32x3x224x224, fp32) in one batch, PyTorch CPU oracle, %d threads; "
                           "1 clip on 1 thread: %.3f clips/s" % (x.shape[0], threads_all, 1.0 / t_one)}


def pmc_traffic(kernel_label, dtype):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_<dtype>_traffic.json, written by tools/summarize_profile.py); None if no such profile exists."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_traffic.json" % dtype)))
    if not files:
        return None, None
    m = re.search(r"BN=(\d+),BM=(\d+)", kernel_label)
    data = json.load(open(files[-1]))["kernels"]
    tot, n = 0.0, 0                      # launch-weighted over every instantiation that carries this label
    for name, v in data.items():
        hit = (m and re.search(r"conv_igemm_kernel<\d+, %s, %s," % (m.group(1), m.group(2)), name)) or \
              (not m and kernel_label.split("<")[0].replace("_kernel", "").replace("_c64", "") + "_" in name)
        if hit:
            tot += v["hbm_bytes_per_launch"] * v["launches_profiled"]
            n += v["launches_profiled"]
    if n:
        return tot / n, os.path.relpath(files[-1], ROOT)
    return None, None


def bench_dualrun(args, rank, world, dev):
    """SURVEY 8d config C4's extra branch on its own: the dualrun AU / landmark dual encoder (T = 8 frames per clip,
    d_model 256, 4 layers) on `--batch` clips per GPU; value = clips/s.  Not the BASELINE metric (use --model i3d)."""
    from af_mi355x import dualrun
    sp = dualrun.DualSpec()
    sd = dualrun.dual_synthetic_state_dict(sp, seed=0)
    net = dualrun.DualEncoderAU_LMK(au_dim=sp.au_dim, lmk_dim=sp.lmk_dim, d_model=sp.d_model, depth=sp.depth, heads=sp.heads,
                                    mlp_ratio=float(sp.ff) / sp.d_model, pool_tau=sp.pool_tau)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    B = args.batch
    A, L, lengths = dualrun.synthetic_dual_inputs(B, sp, frames=8, seed=2026 + rank)
    Ad, Ld, ld = A.to(dev), L.to(dev), lengths.to(dev)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.inference_mode():
        for _ in range(args.warmup):
            out = net(Ad, Ld, ld)["bin_logits"]
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(Ad, Ld, ld)["bin_logits"]
        fence()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    line = {"metric": "clips/sec (dualrun AU+LMK encoder, 8 frames)", "value": round(world * B * args.steps / dt, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dualrun DualEncoderAU_LMK forward (AU 36 + LMK 132 features x 8 frames, d_model 256, 4 layers, "
                                   "4 heads, ff 768), batch=%d clips/GPU, synthetic weights / inputs" % B,
                       "global_batch": world * B, "parallelism": "dp%d" % world}}
    if rank == 0 and world == 1 and args.cpu_clips > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import dualrun_oracle
        torch.set_num_threads(16)
        dualrun_oracle.dual_forward(sd, A[:1], L[:1], lengths[:1], heads=sp.heads, tau=sp.pool_tau)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            ref, _ = dualrun_oracle.dual_forward(sd, A, L, lengths, heads=sp.heads, tau=sp.pool_tau)
        tc = (time.perf_counter() - t0) / reps
        line["cpu_baseline"] = {"value": round(B / tc, 2), "unit": "clips/s", "cores": 16, "kind": "port",
                                "sample": "%d clips x %d repetitions, PyTorch CPU oracle, 16 threads" % (B, reps)}
        line["max_abs_logit_err_vs_cpu_fp32"] = float((out.float().cpu() - ref).abs().max())
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_aligner(args, rank, world, dev):
    """SURVEY 8f rank 5 on its own: FasterCropAlignXRay's warps for `--batch` clips of 32 tracked crops (~420x420) -> 224x224
    per GPU; value = clips/s with the crops and the fitted transforms already resident (the warp launches only); the
    host-inclusive rate (numpy fit + pinned upload + launch + sync) is reported next to it.  Not the BASELINE metric."""
    import numpy as np
    from af_mi355x import aligner
    B, size = args.batch, 224
    al = aligner.FasterCropAlignXRay(size)
    clips = [aligner.synthetic_clip(32, seed=2026 + 100 * rank + i) for i in range(B)]
    staged = []
    for infos, crops in clips:
        boxes = np.array([b for _, _, _, b in infos])
        lt = boxes[:, :2].min(0)
        w, h = boxes[:, 2:].max(0) - lt
        diff = boxes[:, :2] - lt[None]
        tfm, _ = aligner.estimate_batch_transform(np.array([l5 for _, l5, _, _ in infos]) + diff[:, None, :], al.std_points)
        dcrops, offs, host = al.stage_crops(crops, dev)
        staged.append((dcrops, offs, [c.shape for c in crops], diff, int(h), int(w), tfm, host))
    out = torch.empty((B, 32, size, size, 3), dtype=torch.uint8, device=dev)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        for i, (dcrops, offs, shapes, diff, h, w, tfm, _) in enumerate(staged):
            al.launch_warps(dcrops, offs, shapes, diff, h, w, tfm, out[i])

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    crop_bytes = sum(int(np.prod(s)) for st in staged for s in st[2])
    alg = crop_bytes + out.numel()                                   # every crop byte once + the aligned clip
    line = {"metric": "clips/sec (aligner: 32 crops -> 224x224)", "value": round(world * B * args.steps / dt, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "FasterCropAlignXRay warps, batch=%d clips/GPU x 32 crops (380-460 px) -> 224x224x3 uint8, "
                                   "crops and fitted transforms resident in HBM" % B,
                       "global_batch": world * B, "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "kernel": "warp_affine_clip_kernel", "launches_per_step": B,
                         "avg_launch_us": round(1e6 * dt / args.steps / B, 2), "achieved": round(alg / (dt / args.steps) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "algorithmic_bytes_per_launch": alg // B,
                         "note": "launch-bound at this size (a clip is ~22 MB): back-to-back launches, wall clock"}}
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        for infos, crops in clips:
            al(infos, crops, device_output=True)
    line["host_inclusive_clips_per_s"] = round(B * reps / (time.perf_counter() - t0), 2)
    if rank == 0 and world == 1 and args.cpu_clips > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import aligner_oracle
        t0 = time.perf_counter()
        _, ref = aligner_oracle.crop_align([(a, b.copy(), c.copy(), d.copy()) for a, b, c, d in clips[0][0]], clips[0][1], size=size)
        tc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(1.0 / tc, 2), "unit": "clips/s", "cores": 1, "kind": "port",
                                "sample": "1 clip (32 crops), numpy restatement of the fit + fixed-point warp, 1 thread"}
        line["bytes_differing_vs_cpu"] = int((out[0].cpu().numpy() != ref).sum())
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu-clips", type=int, default=4, help="clips in the CPU baseline sample (0 = skip)")
    ap.add_argument("--model", default="i3d", choices=["i3d", "slowfast", "ftcn_tt", "dualrun", "aligner"],
                    help="i3d = the i3d_ori plugin (BASELINE metric); slowfast = the two-pathway SlowFast-R50, ftcn_tt = the "
                         "reference's second plugin (next rows of SURVEY 8f)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--layers-json", default=None, help="write per-layer device times / rates to this file")
    args = ap.parse_args()

    import af_mi355x  # noqa: F401
    from af_mi355x import _lib, parallel, synth
    from af_mi355x.classifier import Classifier
    from af_mi355x.engine import TAG_NAMES

    # AF_BENCH_REHEARSAL=1: all ranks share cuda:0 and talk over gloo - lets the N>1 code path be rehearsed on a
    # one-GPU box; never set by the driver (its ranks get one GPU each and RCCL)
    rehearsal = os.environ.get("AF_BENCH_REHEARSAL") == "1"
    rank, local_rank, world = parallel.init(backend="gloo" if rehearsal else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    B = args.batch

    if args.model == "dualrun":
        return bench_dualrun(args, rank, world, dev)
    if args.model == "aligner":
        return bench_aligner(args, rank, world, dev)
    if args.model == "slowfast":
        from af_mi355x.arch import slowfast_r50_spec
        from af_mi355x.classifier import SlowFast8x8
        sd = synth.synthetic_state_dict(slowfast_r50_spec(), seed=0)
        clf = SlowFast8x8(precision=args.dtype)
        clf.load_state_dict(sd)
        clf = clf.to(dev).eval()
        net = clf
    elif args.model == "ftcn_tt":
        from af_mi355x.arch import ftcn_tt_spec
        from af_mi355x.classifier import FtcnTTClassifier
        sd = synth.synthetic_state_dict(ftcn_tt_spec(), seed=0)
        clf = FtcnTTClassifier(precision=args.dtype)
        clf.network.load_state_dict(sd)
        clf = clf.to(dev).eval()
        net = clf.network
    else:
        sd = synth.synthetic_state_dict(seed=0)
        clf = Classifier(precision=args.dtype)
        clf.network.load_state_dict(sd)
        clf = clf.to(dev).eval()
        net = clf.network
    u8 = synth.synthetic_clips_u8(B, seed=2026 + rank, kind="uniform")
    x = synth.normalize_like_callers(u8.to(dev))                       # (B,3,32,224,224) fp32, channels-last strides
    gathered = torch.empty((world * B, 1), dtype=torch.float32, device=dev)

    def step():
        y = clf(x)["final_output"]
        return parallel.gather_logits(y, world * B, gathered)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.inference_mode():
        for _ in range(args.warmup):
            out = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    clips_per_s = world * B * args.steps / dt

    line = {
        "metric": "clips/sec (32x3x224x224)", "value": round(clips_per_s, 2), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s forward, batch=%d clips/GPU of 32x3x224x224, synthetic checkpoint W(0), "
                               "model.forward(clip) on HBM-resident fp32 input"
                               % ({"i3d": "AltFreezing i3d_ori (I3D-R50)", "slowfast": "SlowFast-R50 (alpha 8)",
                                   "ftcn_tt": "FTCN-TT plugin (i3d_temporal_var_fix_dropout_tt_cfg, ftcn_tt.yaml)"}[args.model], B),
                   "global_batch": world * B, "parallelism": "dp%d + all-gather of logits" % world},
    }

    if rank == 0 and world == 1:
        eng = net._engines[(args.dtype, B, (32, 224, 224))]
        total_macs = sum(eng.op_macs)
        line["model_tflops_per_s"] = round(2 * total_macs / B * clips_per_s / 1e12, 2)
        if not args.no_roofline:
            reps = 5
            acc = [0.0] * eng.n_ops
            with torch.inference_mode():
                for _ in range(reps):
                    ms = eng.run_timed()
                    acc = [a + m for a, m in zip(acc, ms)]
            ms = [a / reps for a in acc]
            # per kernel instantiation (conv variants) and per layer class
            per_kernel, per_class = {}, {}
            es = 4 if args.dtype == "f32" else 2
            eng_bytes, kernel_ops = {}, {}
            for i in range(eng.n_ops):
                op = eng.ops[i]
                if op.kind in (_lib.AF_OP_CONV, _lib.AF_OP_STEM, _lib.AF_OP_CONV_DUAL, _lib.AF_OP_STEM_POOL, _lib.AF_OP_TSTEM):
                    cd = op.conv
                    mm = cd.n * cd.to * cd.ho * cd.wo
                    if op.kind == _lib.AF_OP_STEM_POOL:          # only the pooled tensor is written
                        mm = cd.n * cd.to * ((cd.ho - 1) // 2 + 1) * ((cd.wo - 1) // 2 + 1)
                    elif cd.tpool:
                        mm //= (4 if cd.tpool == 2 else 2)
                    eng_bytes[i] = es * (cd.n * cd.t * cd.h * cd.w * cd.cin + mm * cd.cout
                                         + (cd.n * cd.to * cd.ho * cd.wo * cd.cout if op.residual else 0)
                                         + cd.cout * cd.cin * cd.kt * cd.kh * cd.kw)
                    if op.kind == _lib.AF_OP_CONV_DUAL:
                        c2 = op.conv2
                        eng_bytes[i] += es * (c2.n * c2.t * c2.h * c2.w * c2.cin // (c2.sh * c2.sw) + c2.cout * c2.cin)
                cls = TAG_NAMES[op.tag]
                c = per_class.setdefault(cls, {"ms": 0.0, "macs": 0, "launches": 0})
                c["ms"] += ms[i]; c["macs"] += eng.op_macs[i]; c["launches"] += 1
                if op.kind in (_lib.AF_OP_CONV, _lib.AF_OP_CONV_DUAL):
                    import ctypes as C
                    d2 = C.byref(op.conv2) if op.kind == _lib.AF_OP_CONV_DUAL else None
                    kname = _lib.lib.af_conv_variant_name(_lib.lib.af_conv_variant(C.byref(op.conv), d2)).decode()
                elif op.kind in (_lib.AF_OP_STEM, _lib.AF_OP_STEM_POOL, _lib.AF_OP_TSTEM):
                    kname = {_lib.AF_OP_STEM: "stem_kernel", _lib.AF_OP_STEM_POOL: "stem_pool_kernel",
                             _lib.AF_OP_TSTEM: "tstem_kernel"}[op.kind]
                else:
                    continue
                k = per_kernel.setdefault(kname, {"ms": 0.0, "macs": 0, "launches": 0})
                k["ms"] += ms[i]; k["macs"] += eng.op_macs[i]; k["launches"] += 1
                kernel_ops.setdefault(kname, []).append(i)
            if args.layers_json:
                es = 4 if args.dtype == "f32" else 2
                rows = []
                for i in range(eng.n_ops):
                    op = eng.ops[i]
                    row = {"i": i, "name": eng.op_names[i], "class": TAG_NAMES[op.tag], "ms": round(ms[i], 4)}
                    if i in eng_bytes:
                        cd = op.conv
                        m = cd.n * cd.to * cd.ho * cd.wo
                        byts = eng_bytes[i]
                        row.update({"M": m, "N": cd.cout, "K": cd.cin * cd.kt * cd.kh * cd.kw
                                    + (op.conv2.cin if op.kind == _lib.AF_OP_CONV_DUAL else 0),
                                    "tflops": round(2 * eng.op_macs[i] / ms[i] / 1e9, 1),
                                    "alg_GBs": round(byts / ms[i] / 1e6, 0)})
                    rows.append(row)
                with open(args.layers_json, "w") as f:
                    json.dump(rows, f, indent=0)
            dom = max(per_kernel, key=lambda k: per_kernel[k]["ms"])
            d = per_kernel[dom]
            dom_ops = kernel_ops[dom]
            # which roof bounds this kernel: its algorithmic intensity against the ridge of the machine
            alg_bytes_total = sum(eng_bytes[i] for i in dom_ops)
            intensity = 2 * d["macs"] / max(alg_bytes_total, 1)
            ridge = PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
            if intensity >= ridge:
                achieved = 2 * d["macs"] / (d["ms"] * 1e-3) / 1e12
                peak, unit, bound = PEAK_TFLOPS[args.dtype], "TFLOP/s", "mfma"
            else:
                achieved = alg_bytes_total / (d["ms"] * 1e-3) / 1e9
                peak, unit, bound = HBM_PEAK_GBS, "GB/s", "hbm"
            line["roofline"] = {
                "bound": bound, "kernel": dom, "launches_per_step": d["launches"],
                "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                "algorithmic_gflop_per_launch": round(2 * d["macs"] / d["launches"] / 1e9, 3),
                "algorithmic_intensity_flop_per_byte": round(intensity, 1),
                "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4), "traffic": None,
            }
            tr, src = pmc_traffic(dom, args.dtype)
            if tr is not None:
                line["roofline"]["traffic"] = round(tr)
                line["roofline"]["traffic_unit"] = "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE), " + src
                alg_bytes = sum(eng_bytes[i] for i in dom_ops) / max(len(dom_ops), 1)
                line["roofline"]["algorithmic_bytes_per_launch"] = round(alg_bytes)
            line["device_ms_per_step"] = round(sum(ms), 3)
            line["classes"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                                   "tflops": round(2 * v["macs"] / max(v["ms"], 1e-9) / 1e9, 1) if v["macs"] else None}
                               for k, v in sorted(per_class.items(), key=lambda kv: -kv[1]["ms"])}
        if args.cpu_clips > 0:
            n = min(args.cpu_clips, B)
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            ref, cb = cpu_baseline(sd, u8[:n], max(1, min(cores, 16)), args.model)     # the box's CPU share for one GPU is 16
            line["cpu_baseline"] = cb
            line["max_abs_logit_err_vs_cpu_fp32"] = float((out[:n].float().cpu() - ref).abs().max())
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
