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
# what MI355X_MICROARCH.md says the chip SUSTAINS: 6.29 TB/s measured copy bandwidth; 16-bit MFMA loops on random data hold
# ~1.9 GHz of the 2.4 GHz the 2.5 PFLOP/s peak is quoted at (DVFS give-back) -> ~1.9 PFLOP/s; fp32 MFMA measured 155 TFLOP/s
SUSTAINED_TFLOPS = {"bf16": 1900.0, "f16": 1900.0, "f32": 155.0}
HBM_SUSTAINED_GBS = 6290.0


def cpu_baseline(sd, u8, threads_all, model="i3d"):
    """The oracle (PyTorch-CPU restatement of the reference forward, pinned by tests/golden) timed on the
    host cores: a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import i3d_oracle as oracle
    x = oracle.normalize(u8)
    fwd = {"i3d": oracle.forward, "ftcn_tt": oracle.ftcn_forward,
           "slowfast": (lambda s_, x_: oracle.slowfast_forward(s_, x_[:, :, ::8], x_))}[model]
    torch.set_num_threads(threads_all)

    def timed(xs, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            y = fwd(sd, xs)
            ts.append(time.perf_counter() - t0)
        return y, sorted(ts)[len(ts) // 2]

    with torch.no_grad():
        fwd(sd, x[:1])                                              # warm-up (first call pages oneDNN in)
        ref, t_all = timed(x, 3)                                    # median of 3
        _, t_b1 = timed(x[:1], 3)
        torch.set_num_threads(1)                                    # the real-time app pins OMP_NUM_THREADS=1 (af_realtime.py:4-7)
        _, t_one = timed(x[:1], 1)
    torch.set_num_threads(threads_all)
    return ref, {"value": round(x.shape[0] / t_all, 4), "unit": "clips/s", "cores": threads_all, "kind": "port",
                 "batch1_value": round(1.0 / t_b1, 4), "one_thread_value": round(1.0 / t_one, 4),
                 "sample": "PyTorch CPU oracle (restatement of the reference forward, pinned by tests/golden), fp32: "
                           "%d clips (32x3x224x224) in one batch x 3 timed forwards, median, %d threads (= value); "
                           "batch 1 x 3, median, same threads (= batch1_value); 1 clip on 1 thread x 1 (= one_thread_value)"
                           % (x.shape[0], threads_all)}


def kernel_family(kname: str) -> str:
    """conv_igemm<BN=128,BM=256> -> conv_igemm: every tile instantiation of one kernel template is ONE kernel for the
    roofline (rocprof lists them as conv_igemm_kernel<...> instantiations of the same source)."""
    return kname.split("<")[0]


def pmc_traffic(family, dtype):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes of this same command
    (profiles/*_<dtype>_traffic.json, tools/summarize_profile.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950's wide streaming reads); launch-weighted over the family's instantiations."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_traffic.json" % dtype)))
    if not files:
        return None, None, None
    data = json.load(open(files[-1]))["kernels"]
    key = {"conv_igemm": "conv_igemm_kernel<", "conv133_c64": "conv133_c64_kernel<", "conv311_c64": "conv311_kernel<",
           "conv111": "conv111_kernel<", "conv_small": "conv_small_kernel<", "stem_pool_kernel": "stem_pool_kernel<",
           "stem_kernel": "stem_kernel<", "tstem_kernel": "tstem_kernel<", "conv133": "conv133_kernel<"}.get(family, family + "<")
    tot = raw = 0.0
    n = 0
    for name, v in data.items():
        if key in name:
            tot += v["hbm_bytes_per_launch"] * v["launches_profiled"]
            raw += (v.get("fetch_size_raw_bytes_per_launch", v["hbm_read_bytes_per_launch"] / 2) +
                    v["hbm_write_bytes_per_launch"]) * v["launches_profiled"]
            n += v["launches_profiled"]
    if n:
        return tot / n, raw / n, os.path.relpath(files[-1], ROOT)
    return None, None, None


def roofline_report(eng, args, line, reps=5):
    """Per-op device time from hipEvents on the launch stream (af_run_ops_timed), grouped (a) by kernel family - all tile
    instantiations of conv_igemm are one kernel - and (b) by layer class.  `roofline` = the family with the most device
    time, priced against the roof its algorithmic intensity puts it under; `mfma_roofline` = achieved / dense-MFMA peak for
    every conv class and the whole model (the "fraction of the Conv3d MFMA roofline" BASELINE.json asks for)."""
    import ctypes as C
    from af_mi355x import _lib
    from af_mi355x.engine import TAG_NAMES
    acc = [0.0] * eng.n_ops
    with torch.inference_mode():
        for _ in range(reps):
            acc = [a + m for a, m in zip(acc, eng.run_timed())]
    ms = [a / reps for a in acc]
    es = 4 if args.dtype == "f32" else 2
    per_kernel, per_class, eng_bytes, kernel_ops, variants, op_kernel = {}, {}, {}, {}, {}, {}
    for i in range(eng.n_ops):
        op = eng.ops[i]
        if op.kind == _lib.AF_OP_CONV_CA:                # b tile + residual + trunk out + a out + both weights; the trunk is not re-read
            cc_, ca = op.conv, op.conv2
            pos = cc_.n * cc_.to * cc_.ho * cc_.wo
            # plain block: b tile + residual + trunk out + a out; projection block: b tile + shortcut input + trunk out + a out
            side = op.conv3.cin if op.in3 else cc_.cout
            eng_bytes[i] = es * (pos * (cc_.cin + side + cc_.cout + ca.cout) + cc_.cout * (cc_.cin + (op.conv3.cin if op.in3 else 0))
                                 + ca.cout * ca.cin * 3)
        elif op.kind == _lib.AF_OP_CONV_CPA:             # b tile + residual in; pooled trunk (whole or 1 position in 4) + a out
            cc_, ca = op.conv, op.conv2
            pos, posp = cc_.n * cc_.to * cc_.ho * cc_.wo, ca.n * ca.to * ca.ho * ca.wo
            eng_bytes[i] = es * (pos * (cc_.cin + cc_.cout) + posp * cc_.cout // (op.x_sub * op.x_sub) + posp * ca.cout
                                 + cc_.cout * cc_.cin + ca.cout * ca.cin * 3)
        elif op.kind == _lib.AF_OP_BLOCK_ABC:            # trunk in + trunk out + the three weights; a and b stay on chip
            ca, cb, cc_ = op.conv, op.conv2, op.conv3
            pos = ca.n * ca.t * ca.h * ca.w
            eng_bytes[i] = es * (pos * (ca.cin + cc_.cout) + ca.cout * ca.cin * ca.kt + cb.cout * cb.cin * 9 + cc_.cout * cc_.cin
                                 + (op.conv4.cout * op.conv4.cin if op.weight4 else 0))
        elif op.kind == _lib.AF_OP_CONV_BC:              # b input + c output + residual + both weights; the b output stays on chip
            cb, cc_ = op.conv, op.conv2
            pos = cb.n * cb.to * cb.ho * cb.wo
            eng_bytes[i] = es * (pos * cb.cin + pos * cc_.cout * (2 if op.residual else 1) + cb.cout * cb.cin * 9 + cc_.cout * cc_.cin)
        elif op.kind in (_lib.AF_OP_CONV, _lib.AF_OP_STEM, _lib.AF_OP_CONV_DUAL, _lib.AF_OP_STEM_POOL, _lib.AF_OP_STEM3_POOL, _lib.AF_OP_TSTEM, _lib.AF_OP_TSTEM_POOL3):
            cd = op.conv
            mm = cd.n * cd.to * cd.ho * cd.wo
            if op.kind in (_lib.AF_OP_STEM_POOL, _lib.AF_OP_STEM3_POOL, _lib.AF_OP_TSTEM_POOL3):          # only the pooled tensor is written
                mm = cd.n * cd.to * ((cd.ho - 1) // 2 + 1) * ((cd.wo - 1) // 2 + 1)
            elif cd.tpool:
                mm //= (4 if cd.tpool == 2 else 2)
            # algorithmic bytes of a launch: input + output (+ residual) + weights, each once
            eng_bytes[i] = es * (cd.n * cd.t * cd.h * cd.w * cd.cin + mm * cd.cout
                                 + (cd.n * cd.to * cd.ho * cd.wo * cd.cout if op.residual else 0)
                                 + cd.cout * cd.cin * cd.kt * cd.kh * cd.kw)
            if op.kind == _lib.AF_OP_CONV_DUAL:
                c2 = op.conv2
                eng_bytes[i] += es * (c2.n * c2.t * c2.h * c2.w * c2.cin // (c2.sh * c2.sw) + c2.cout * c2.cin)
        elif op.kind in (_lib.AF_OP_PACK_F32, _lib.AF_OP_PACK3_F32, _lib.AF_OP_PACK_U8, _lib.AF_OP_PACK3_U8):
            cd = op.conv                                 # caller's clip (fp32 or uint8, 3 channels) read once + the padded stem input written
            src_es = 1 if op.kind in (_lib.AF_OP_PACK_U8, _lib.AF_OP_PACK3_U8) else 4
            eng_bytes[i] = cd.n * cd.t * cd.h * cd.w * 3 * src_es + eng.buf[eng.op_dst[i]].numel() * es
        elif op.kind == _lib.AF_OP_HEAD:
            pd = op.pool
            eng_bytes[i] = es * pd.n * pd.t * pd.h * pd.w * pd.c
        c = per_class.setdefault(TAG_NAMES[op.tag], {"ms": 0.0, "macs": 0, "launches": 0, "bytes": 0})
        c["ms"] += ms[i]; c["macs"] += eng.op_macs[i]; c["launches"] += 1; c["bytes"] += eng_bytes.get(i, 0)
        if op.kind in (_lib.AF_OP_CONV, _lib.AF_OP_CONV_DUAL):
            d2 = C.byref(op.conv2) if op.kind == _lib.AF_OP_CONV_DUAL else None
            kname = _lib.lib.af_conv_variant_name(_lib.lib.af_conv_variant(C.byref(op.conv), d2)).decode()
        elif op.kind == _lib.AF_OP_CONV_CA:
            kname = "conv_ca<c(i) -> a(i+1) fused>"
        elif op.kind == _lib.AF_OP_CONV_CPA:
            kname = "conv_cpa<c + temporal pool -> a of the next stage>"
        elif op.kind == _lib.AF_OP_CONV_BC:
            kname = "conv133g<b + c fused>"
        elif op.kind == _lib.AF_OP_BLOCK_ABC:
            kname = "block_abc<a + b + c of a narrow block>"
        elif op.kind in (_lib.AF_OP_STEM, _lib.AF_OP_STEM_POOL, _lib.AF_OP_STEM3_POOL, _lib.AF_OP_TSTEM, _lib.AF_OP_TSTEM_POOL3):
            kname = {_lib.AF_OP_STEM: "stem_kernel", _lib.AF_OP_STEM_POOL: "stem_pool_kernel", _lib.AF_OP_STEM3_POOL: "stem3_pool_kernel",
                     _lib.AF_OP_TSTEM: "tstem_kernel", _lib.AF_OP_TSTEM_POOL3: "tstem_pool3_kernel"}[op.kind]
        else:
            continue
        op_kernel[i] = kname
        fam = kernel_family(kname)
        k = per_kernel.setdefault(fam, {"ms": 0.0, "macs": 0, "launches": 0})
        k["ms"] += ms[i]; k["macs"] += eng.op_macs[i]; k["launches"] += 1
        kernel_ops.setdefault(fam, []).append(i)
        v = variants.setdefault(fam, {}).setdefault(kname, {"ms": 0.0, "launches": 0})
        v["ms"] += ms[i]; v["launches"] += 1
    # speed of light per launch: the longer of its algorithmic FLOP at the MFMA peak and its algorithmic bytes at the HBM peak
    # (sol_ms), and the same at what the chip sustains (sol_sustained_ms); gap_ms = measured - sol_ms is the work queue
    peak_tf0, sus_tf0 = PEAK_TFLOPS[args.dtype], SUSTAINED_TFLOPS[args.dtype]
    sol = [max(2 * eng.op_macs[i] / (peak_tf0 * 1e12), eng_bytes.get(i, 0) / (HBM_PEAK_GBS * 1e9)) * 1e3 for i in range(eng.n_ops)]
    sol_s = [max(2 * eng.op_macs[i] / (sus_tf0 * 1e12), eng_bytes.get(i, 0) / (HBM_SUSTAINED_GBS * 1e9)) * 1e3 for i in range(eng.n_ops)]
    unpriced = [eng.op_names[i] for i in range(eng.n_ops) if i not in eng_bytes and not eng.op_macs[i]]
    gaps = sorted(range(eng.n_ops), key=lambda i: sol[i] - ms[i])
    line["speed_of_light"] = {
        "ms": round(sum(sol), 4), "frac": round(sum(sol) / sum(ms), 4),
        "sustained_ms": round(sum(sol_s), 4), "sustained_frac": round(sum(sol_s) / sum(ms), 4),
        "definition": "per launch max(algorithmic FLOP / %.0f TFLOP/s, algorithmic bytes / %.0f GB/s), summed over the %d launches "
                      "of a step; sustained = the same at %.0f TFLOP/s and %.0f GB/s (MI355X_MICROARCH.md: DVFS clock under MFMA "
                      "load, measured copy bandwidth); frac = that sum / measured device ms per step"
                      % (peak_tf0, HBM_PEAK_GBS, eng.n_ops, sus_tf0, HBM_SUSTAINED_GBS),
        "largest_gaps": [{"i": i, "name": eng.op_names[i], "ms": round(ms[i], 4), "sol_ms": round(sol[i], 4),
                          "gap_ms": round(ms[i] - sol[i], 4)} for i in gaps[:8]],
        "unpriced_launches": unpriced}
    if args.layers_json:
        rows = []
        for i in range(eng.n_ops):
            op = eng.ops[i]
            row = {"i": i, "name": eng.op_names[i], "class": TAG_NAMES[op.tag], "ms": round(ms[i], 4),
                   "sol_ms": round(sol[i], 4), "sol_sustained_ms": round(sol_s[i], 4), "gap_ms": round(ms[i] - sol[i], 4),
                   "alg_bytes": int(eng_bytes.get(i, 0)), "alg_flop": int(2 * eng.op_macs[i])}
            if i in op_kernel:
                row["kernel"] = op_kernel[i]
            if i in eng_bytes and op.kind not in (_lib.AF_OP_PACK_F32, _lib.AF_OP_PACK3_F32, _lib.AF_OP_PACK_U8, _lib.AF_OP_PACK3_U8, _lib.AF_OP_HEAD):
                cd = op.conv
                row.update({"M": cd.n * cd.to * cd.ho * cd.wo, "N": op.conv2.cout if op.kind == _lib.AF_OP_CONV_BC else cd.cout,
                            "K": cd.cin * cd.kt * cd.kh * cd.kw + (op.conv2.cin if op.kind in (_lib.AF_OP_CONV_DUAL, _lib.AF_OP_CONV_BC) else 0),
                            "tflops": round(2 * eng.op_macs[i] / ms[i] / 1e9, 1),
                            "alg_GBs": round(eng_bytes[i] / ms[i] / 1e6, 0)})
            rows.append(row)
        with open(args.layers_json, "w") as f:
            json.dump(rows, f, indent=0)
    peak_tf = PEAK_TFLOPS[args.dtype]
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)

    def priced(macs, byts, t_ms):
        """(bound, achieved, peak, unit) of a group of launches: MFMA roof when its algorithmic intensity is above the ridge"""
        if 2 * macs / max(byts, 1) >= ridge:
            return "mfma", 2 * macs / (t_ms * 1e-3) / 1e12, peak_tf, "TFLOP/s"
        return "hbm", byts / (t_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"

    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms"])
    d, dom_ops = per_kernel[dom], kernel_ops[dom]
    alg_bytes_total = sum(eng_bytes[i] for i in dom_ops)
    bound, achieved, peak, unit = priced(d["macs"], alg_bytes_total, d["ms"])
    line["roofline"] = {
        "bound": bound, "kernel": dom + "_kernel" if not dom.endswith("_kernel") else dom,
        "instantiations": {k: {"launches": v["launches"], "ms": round(v["ms"], 3)} for k, v in variants[dom].items()},
        "launches_per_step": d["launches"], "share_of_device_time": round(d["ms"] / sum(ms), 3),
        "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
        "algorithmic_gflop_per_launch": round(2 * d["macs"] / d["launches"] / 1e9, 3),
        "algorithmic_bytes_per_launch": round(alg_bytes_total / d["launches"]),
        "algorithmic_intensity_flop_per_byte": round(2 * d["macs"] / max(alg_bytes_total, 1), 1),
        "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4), "traffic": None,
    }
    tr, raw, src = pmc_traffic(dom, args.dtype)
    if tr is not None:
        line["roofline"]["traffic"] = round(tr)
        line["roofline"]["traffic_uncorrected"] = round(raw)
        line["roofline"]["traffic_unit"] = ("HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction; "
                                            "traffic_uncorrected = FETCH_SIZE + WRITE_SIZE as counted), " + src)
        line["roofline"]["traffic_source"] = ("committed profile (%s: separate rocprofv3 --pmc passes of this same command), NOT measured "
                                              "in this run - PMC counters cannot be read from inside the timed process" % src)
    # the fraction of the Conv3d MFMA roofline, per conv class and for the whole model (all launches, pack and head included)
    mf = {}
    for cls, v in per_class.items():
        if v["macs"]:
            tf = 2 * v["macs"] / (v["ms"] * 1e-3) / 1e12
            mf[cls] = {"achieved": round(tf, 1), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4),
                       "ms": round(v["ms"], 3), "launches": v["launches"],
                       "bound_by_intensity": "mfma" if 2 * v["macs"] / max(v["bytes"], 1) >= ridge else "hbm",
                       "algorithmic_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9)}
    tot_macs = sum(eng.op_macs)
    tf = 2 * tot_macs / (sum(ms) * 1e-3) / 1e12
    mf["model"] = {"achieved": round(tf, 1), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4),
                   "ms": round(sum(ms), 3), "launches": eng.n_ops}
    line["mfma_roofline"] = mf
    line["kernels"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                           "tflops": round(2 * v["macs"] / max(v["ms"], 1e-9) / 1e9, 1)}
                       for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms"])}
    line["device_ms_per_step"] = round(sum(ms), 3)
    line["classes"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                           "tflops": round(2 * v["macs"] / max(v["ms"], 1e-9) / 1e9, 1) if v["macs"] else None}
                       for k, v in sorted(per_class.items(), key=lambda kv: -kv[1]["ms"])}


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


def bench_dualrun_rgb(args, rank, world, dev):
    """BASELINE config[3]: "dualrun two-stream (rgb + temporal) model, batch=16 clips, 1xMI355X" as ONE step on the device:
    AltFreezing forward of `--batch` uint8 clips (32x224x224, fused normalise prologue) -> its pooled 2048-vector (the RGB stream's
    feature, feature.py:105-114) -> the tri-modal DualEncoderRGB (dualrun/model/dual_rgb.py: AU / landmark branch encoders over
    8 frames + masked-mean RGB feature + rgb_proj + head) -> GatedMoE late fusion with the AltFreezing logit (rgb/engine_rgb.py:369-404).
    value = clips/s.  Weights / AU / landmark tracks are synthetic (trained checkpoints and the LibreFace / MediaPipe extractors are
    not available offline)."""
    from af_mi355x import dualrun, synth
    from af_mi355x.classifier import Classifier
    B = args.batch
    sdc = synth.synthetic_state_dict(seed=0)
    clf = Classifier(precision=args.dtype)
    clf.network.load_state_dict(sdc)
    clf = clf.to(dev).eval()
    sp = dualrun.DualSpec(36, 132, 256, 4, 4, 768, 0.7, 128)
    sdd = dualrun.dual_rgb_synthetic_state_dict(sp, 2048, seed=0)
    net = dualrun.DualEncoderRGB(36, 132, 2048, d_model=256, depth=4, heads=4, ff_dim=3.0, rgb_backbone=clf, rgb_from_features=False)
    net.load_state_dict(sdd)
    net = net.to(dev).eval()
    moe = dualrun.GatedMoE().to(dev).eval()
    u8 = synth.synthetic_clips_u8(B, seed=2026 + rank, kind="uniform")
    A, L, lengths = dualrun.synthetic_dual_inputs(B, sp, frames=8, seed=2026 + rank)
    ud, Ad, Ld = u8.to(dev), A.to(dev), L.to(dev)
    mask = net.lengths_to_mask(lengths, 8, dev)
    ms = {"altfreezing": 0.0, "dual_rgb": 0.0}

    def step():
        # the module runs its RGB backbone itself (rgb_from_features=False): one AltFreezing forward gives the RGB logit and the
        # pooled feature V, and the AU / landmark branch encoders run beside it on a side stream
        z_dual, rgb = net(Ad, Ld, ud, key_padding_mask=mask, return_rgb=True)
        return moe(rgb["final_output"], z_dual.view(B, 1)), rgb, z_dual

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.inference_mode():
        for _ in range(args.warmup):
            (z, gate), rgb, z_dual = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            (z, gate), rgb, z_dual = step()
        fence()
        dt = time.perf_counter() - t0
        # share of the AltFreezing forward in a step (events on the launch stream)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        net.rgb_from_features = True
        try:
            e[0].record(); rgb2 = clf.network.forward_clips_u8(ud, return_pooled=True); e[1].record()
            net(Ad, Ld, rgb2["pooled"].view(B, 1, -1), key_padding_mask=mask); e[2].record()
        finally:
            net.rgb_from_features = False
        torch.cuda.synchronize(dev)
        ms["altfreezing"], ms["dual_rgb"] = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    line = {"metric": "clips/sec (dualrun two-stream: AltFreezing 32x3x224x224 + AU/LMK 8 frames)", "value": round(world * B * args.steps / dt, 2),
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype + " (AltFreezing trunk) / f32 (dual encoder, head, MoE)",
            "data": "synthetic",
            "config": {"workload": "BASELINE config[3]: AltFreezing i3d_ori forward (uint8 clips, fused prologue) -> pooled 2048-d feature -> "
                                   "DualEncoderRGB (AU 36 + LMK 132 x 8 frames, d_model 256, 4 layers, ff 768; rgb_proj; 3d head) -> GatedMoE, "
                                   "batch=%d clips/GPU, synthetic weights / tracks" % B,
                       "global_batch": world * B, "parallelism": "dp%d" % world},
            "stage_ms": {k: round(v, 3) for k, v in ms.items()},
            "stage_ms_note": "the two stages timed back to back on one stream; in the step the AU / landmark encoders run beside the AltFreezing forward on a side stream"}
    if rank == 0 and world == 1 and args.cpu_clips > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import dualrun_oracle
        import i3d_oracle
        n = min(args.cpu_clips, B)
        torch.set_num_threads(16)
        x = i3d_oracle.normalize(u8[:n])
        i3d_oracle.forward(sdc, x[:1])
        t0 = time.perf_counter()
        ref_logit, stages = i3d_oracle.forward(sdc, x, return_stages=True)
        feat = stages["avgpool"].reshape(n, 1, -1)
        ref_dual, _ = dualrun_oracle.dual_rgb_forward(sdd, A[:n], L[:n], feat, dualrun_oracle.lengths_to_mask(lengths[:n], 8), heads=4, tau=0.7)
        msd = {k: v.detach().cpu() for k, v in moe.state_dict().items()}
        ref_z, _ = dualrun_oracle.gated_moe(msd, ref_logit, ref_dual.view(n, 1))
        tc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(n / tc, 3), "unit": "clips/s", "cores": 16, "kind": "port",
                                "sample": "%d clips, PyTorch CPU oracles (AltFreezing + DualEncoderRGB + GatedMoE restatements), fp32, 16 threads, 1 timed pass" % n}
        line["max_abs_err_vs_cpu_fp32"] = {"altfreezing_logit": float((rgb["final_output"][:n].cpu() - ref_logit).abs().max()),
                                           "dual_logit": float((z_dual[:n].cpu() - ref_dual).abs().max()),
                                           "fused_logit": float((z[:n].cpu() - ref_z).abs().max())}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_stream(args, rank, world, dev):
    """BASELINE config[4] stand-in: the af_realtime.py loop around the hot path on a synthetic 1080p30 stream with ONE
    tracked face.  The detectors / tracker (YuNet, ByteTrack: cv2, lap, cython_bbox - absent offline) are out of scope, so the
    stream is pre-synthesised as what they hand over: per window 32 tracked crops (~420 px) + 5 / 68-point landmarks.  A window of
    clip_size = 32 frames closes every stride = 30 frames (test/app_realtime.py:153), i.e. once per second at 30 fps; at that
    moment it is "enqueued" and goes through aligner (FasterCropAlignXRay, device output) -> uint8 prologue -> AltFreezing forward
    (B = 1) -> sigmoid -> host.  Reported: enqueue -> score latency p50 / p95 under real-time pacing (the GPU idles between windows,
    as in a live call), and the sustained rate of the same pipeline back to back (= how many 30-fps tracks one GPU keeps up with)."""
    import numpy as np
    from af_mi355x import aligner, synth
    from af_mi355x.classifier import Classifier
    sd = synth.synthetic_state_dict(seed=0)
    clf = Classifier(precision=args.dtype)
    clf.network.load_state_dict(sd)
    clf = clf.to(dev).eval()
    al = aligner.FasterCropAlignXRay(224, device=dev)
    windows = [aligner.synthetic_clip(32, seed=2026 + 100 * rank + i) for i in range(4)]
    fps, stride = 30.0, 30
    period = stride / fps

    def process(k):
        infos, crops = windows[k % len(windows)]
        _, clip = al(infos, crops, device_output=True)                   # (32, 224, 224, 3) uint8 in HBM
        return clf.network.infer_scores(clip.unsqueeze(0))               # numpy (1,): synchronises

    # the live form (round 3): a crop goes to the GPU when its frame is captured (StreamingCropAligner.push, one small async H2D
    # per frame); closing a window = the last frame's push + the fit + ONE warp launch over resident crops + forward + score
    sal = aligner.StreamingCropAligner(224, capacity=64, device=dev)

    def capture(k, frames):
        infos, crops = windows[k % len(windows)]
        for i in frames:
            sal.push(infos[i], crops[i])

    from af_mi355x.classifier import LiveScorer
    scorer = LiveScorer(clf.network)                                     # the B = 1 forward as one graph replay

    def close_window(k):
        capture(k, (31,))                                                # the frame that closes the window
        sal.align_last(32, out=scorer.clip[0])                           # the warp writes the forward's static input
        return scorer()

    def paced(fn_before, fn_timed):
        out = []
        t0 = time.perf_counter()
        for k in range(args.steps):                                      # real-time pacing: window k closes at t0 + (k + 1) * period
            if fn_before is not None:
                fn_before(k)                                             # frames captured while the window is open
            due = t0 + (k + 1) * period * args.pace
            while True:
                now = time.perf_counter()
                if now >= due:
                    break
                time.sleep(min(0.005, due - now))
            ts = time.perf_counter()
            sc = fn_timed(k)
            out.append(1e3 * (time.perf_counter() - ts))
        return np.array(out), sc

    def paced_live():
        # the live form at the camera's cadence: frame i of window k is pushed at its capture time (one frame every 1 / fps s, the
        # host blocked in between as a capture loop is in cap.read()); the window's last frame -> score is timed
        out = []
        t0 = time.perf_counter()
        frame_t = period * args.pace / 32.0
        for k in range(args.steps):
            for i in range(32):
                due = t0 + (k * 32 + i + 1) * frame_t
                while True:
                    now = time.perf_counter()
                    if now >= due:
                        break
                    time.sleep(min(0.005, due - now))
                if i < 31:
                    capture(k, (i,))
                else:
                    ts = time.perf_counter()
                    sc = close_window(k)
                    out.append(1e3 * (time.perf_counter() - ts))
        return np.array(out), sc

    with torch.inference_mode():
        for k in range(max(args.warmup, 2)):
            process(k)
            capture(k, range(31)); close_window(k)
        torch.cuda.synchronize(dev)
        lat_batch, s_batch = paced(None, process)
        lat, s = paced_live()
        n_sus = 50
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for k in range(n_sus):
            capture(k, range(31))
            s = close_window(k)
        sustained = n_sus / (time.perf_counter() - t1)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for k in range(n_sus):
            s_batch = process(k)
        sustained_batch = n_sus / (time.perf_counter() - t1)
    line = {"metric": "enqueue->score latency, 1080p30 stream stand-in (1 track, 32-frame windows every 30 frames)",
            "value": round(float(np.percentile(lat, 50)), 3), "unit": "ms (p50)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(float(lat.mean()), 3), "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE config[4] stand-in: per window 32 tracked crops (~420 px, uploaded as their frames arrive) -> similarity fit + HIP warp -> "
                                   "uint8 prologue -> AltFreezing i3d_ori forward B=1 -> sigmoid -> host; windows paced at %.2f s (x%.2f)"
                                   % (period, args.pace), "clip_size": 32, "stride_frames": stride, "fps": fps},
            "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "p95": round(float(np.percentile(lat, 95)), 3),
                           "max": round(float(lat.max()), 3), "windows": int(lat.size)},
            "mode": "live: crops uploaded per captured frame at the 30-fps cadence (StreamingCropAligner), forward replayed from one HIP graph "
                    "(LiveScorer); the window's last frame -> score is timed",
            "latency_ms_all_crops_at_window_close": {"p50": round(float(np.percentile(lat_batch, 50)), 3),
                                                     "p95": round(float(np.percentile(lat_batch, 95)), 3), "max": round(float(lat_batch.max()), 3),
                                                     "sustained_clips_per_s": round(sustained_batch, 2),
                                                     "note": "the reference's shape of the call: FasterCropAlignXRay(all 32 crops) when the window closes"},
            "score_difference_between_modes": abs(float(s[0]) - float(s_batch[0])),
            "sustained_clips_per_s": round(sustained, 2),
            "tracks_at_30fps_per_gpu": round(sustained * period, 1),
            "last_score": float(s[0]),
            "not_measured": "face detection / tracking / landmarks on the 1080p frames (cv2, MediaPipe, lap absent offline)"}
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


def conv3x3x3_shapes(args, rank, dev, steps=None):
    """[per-shape record] of the synthetic 3x3x3 Conv3d + BN + ReLU launches (see bench_conv3x3x3), device time from events on
    the launch stream; also called from the default --model i3d line (`conv3x3x3` sub-record)."""
    steps = steps or args.steps
    import ctypes as C
    import torch.nn.functional as F
    from af_mi355x import _lib
    lib, code = _lib.lib, _lib.DTYPE_CODES[args.dtype]
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    es = 4 if args.dtype == "f32" else 2
    B = args.batch
    g = torch.Generator().manual_seed(2026 + rank)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    shapes = []
    for tag, (cin, cout, t, hw) in (("#4 64->64 @32x56x56", (64, 64, 32, 56)), ("#18 256->256 @16x14x14", (256, 256, 16, 14))):
        x = (torch.randn((B, t, hw, hw, cin), generator=g)).to(tdt)                    # NDHWC, rounded once to the compute dtype
        w = torch.randn((cout, cin, 3, 3, 3), generator=g) * (2.0 / (27 * cin)) ** 0.5
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        d = _lib.ConvDesc()
        d.n, d.t, d.h, d.w, d.cin, d.cout = B, t, hw, hw, cin, cout
        d.kt = d.kh = d.kw = 3
        d.st = d.sh = d.sw = d.pt = d.ph = d.pw = 1
        d.to, d.ho, d.wo, d.relu, d.dtype = t, hw, hw, 1, code
        xd, wd = x.to(dev), w.to(dev).contiguous()
        cpad = lib.af_padded_channels(cout)
        sc, sf = torch.zeros(cpad, device=dev), torch.zeros(cpad, device=dev)
        sc[:cout], sf[:cout] = scale.to(dev), shift.to(dev)
        packed = torch.empty(lib.af_packed_conv_weight_bytes(cout, cin, 3, 3, 3, code) // es, dtype=tdt, device=dev)
        _lib.check(lib.af_pack_conv_weight(wd.data_ptr(), cout, cin, 3, 3, 3, code, packed.data_ptr(), st), "pack")
        out = torch.empty((B, t, hw, hw, cout), dtype=tdt, device=dev)

        def launch():
            _lib.check(lib.af_conv3d_bn_act(C.byref(d), xd.data_ptr(), packed.data_ptr(), sc.data_ptr(), sf.data_ptr(), None,
                                            out.data_ptr(), 0, None, 0, st), "conv3x3x3")
        for _ in range(max(args.warmup, 3)):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(steps):
            launch()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        macs = B * t * hw * hw * cout * cin * 27
        # parity: clip 0 against F.conv3d in fp32 on the SAME rounded operands (weights rounded to the compute dtype as the
        # packer does), so the difference is accumulation order + the output rounding only
        x0 = x[:1].float().permute(0, 4, 1, 2, 3)
        ref = F.relu(F.conv3d(x0, w.to(tdt).float(), None, 1, 1) * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1))
        got = out[:1].float().cpu().permute(0, 4, 1, 2, 3)
        rel = float((got - ref).abs().max() / ref.abs().max())
        tf = 2 * macs / (ms * 1e-3) / 1e12
        shapes.append({"shape": tag, "gemm": "M=%d N=%d K=%d" % (B * t * hw * hw, cout, 27 * cin),
                       "kernel": lib.af_conv_variant_name(lib.af_conv_variant(C.byref(d), None)).decode(),
                       "us_per_launch": round(1e3 * ms, 2), "gflop_per_launch": round(2 * macs / 1e9, 2),
                       "achieved": round(tf, 1), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                       "frac": round(tf / PEAK_TFLOPS[args.dtype], 4), "max_rel_err_vs_F_conv3d_fp32": rel})
        del x, xd, out
    return shapes


def bench_conv3x3x3(args, rank, world, dev):
    """The literal "MFMA % on 3x3x3 Conv3d" of BASELINE.json's metric.  SYNTHETIC - NOT A LAYER OF THE REFERENCE MODEL (its
    bottleneck is factorised into 3x1x1 + 1x3x3): the generic kT x kH x kW implicit-GEMM kernel with a full
    3x3x3 kernel (round 3: the frame-resident halo kernel conv133g with kT = 3) on the geometry of SURVEY 8d shape #4 (64 -> 64 @ 32x56x56, K = 1728) and #18 (256 -> 256 @ 16x14x14,
    K = 6912), batch 16, Conv3d + BN + ReLU in one launch; parity against F.conv3d (fp32, CPU) on clip 0."""
    B = args.batch
    shapes = conv3x3x3_shapes(args, rank, dev)
    tot_flop = sum(s["gflop_per_launch"] for s in shapes)
    tot_ms = sum(s["us_per_launch"] for s in shapes) / 1e3
    line = {"metric": "MFMA % on 3x3x3 Conv3d (synthetic: not a layer of the reference model)",
            "value": round(100 * tot_flop / tot_ms / PEAK_TFLOPS[args.dtype], 2), "unit": "%% of dense %s MFMA peak (%.0f TFLOP/s)"
            % (args.dtype, PEAK_TFLOPS[args.dtype]), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(tot_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "Conv3d 3x3x3 (stride 1, pad 1) + BN + ReLU, batch=%d, af_conv3d_bn_act (kernel per shape in "
                                   "`shapes`), SURVEY 8d shapes #4 and #18 with kT=3" % B},
            "shapes": shapes}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n_gpus: int, argv=None, poll_s: float = 0.2) -> int:
    """`python bench.py --gpus N` with no launcher: start N fresh rank processes of this script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them), relay rank 0's JSON line,
    return non-zero if any rank fails.  Runs BEFORE anything touches the GPU: the parent never initialises HIP and
    never execs - the ranks are ordinary children.  All children are polled: as soon as one exits non-zero the others are
    terminated (they would otherwise sit in the rendezvous or in a collective until the process-group timeout), the tail
    of the failing rank's stderr is relayed and the parent returns 1 within seconds.  A rendezvous-port collision (the
    port is found by bind(0) + close, so another process can take it before rank 0 binds) is retried once."""
    import socket
    import subprocess
    import tempfile
    argv = sys.argv[1:] if argv is None else argv

    def attempt():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs, errs = [], []
        for r in range(n_gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            errs.append(tempfile.TemporaryFile())
            out = tempfile.TemporaryFile() if r == 0 else subprocess.DEVNULL
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                           stdout=out, stderr=errs[-1]), out))
        failed = None
        while failed is None and any(p.poll() is None for p, _ in procs):
            for r, (p, _) in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = r
                    break
            else:
                time.sleep(poll_s)
        if failed is None:
            failed = next((r for r, (p, _) in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:                                   # fresh children only: plain terminate, then kill
            for p, _ in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 5.0
            for p, _ in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        tails = []
        for f in errs:
            f.seek(0)
            tails.append(f.read().decode(errors="replace"))
            f.close()
        procs[0][1].seek(0)
        out0 = procs[0][1].read().decode(errors="replace")
        procs[0][1].close()
        return failed, [p.returncode for p, _ in procs], out0, tails

    for tries in range(2):
        failed, codes, out0, tails = attempt()
        if failed is not None and tries == 0 and any("EADDRINUSE" in t or "Address already in use" in t for t in tails):
            continue
        break
    if failed is None:
        sys.stdout.write(out0)
        sys.stdout.flush()
        return 0
    sys.stderr.write("bench.py: rank %d exited with code %s (all exit codes: %s); the other ranks were terminated\n"
                     "---- rank %d stderr (tail) ----\n%s\n" % (failed, codes[failed], codes, failed, tails[failed][-3000:]))
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu-clips", type=int, default=4, help="clips in the CPU baseline sample (0 = skip; 16 = the B=16 figure of SURVEY 8d, ~1 min of CPU time: the default 4 keeps the driver's run short)")
    ap.add_argument("--model", default="i3d", choices=["i3d", "slowfast", "ftcn_tt", "dualrun", "dualrun_rgb", "aligner", "conv3x3x3", "stream"],
                    help="i3d = the i3d_ori plugin (BASELINE metric); slowfast = the two-pathway SlowFast-R50, ftcn_tt = the "
                         "reference's second plugin (next rows of SURVEY 8f)")
    ap.add_argument("--pace", type=float, default=1.0, help="--model stream: multiplier on the real-time window period (1.0 = 30 fps)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the f16 (tolerance-meeting) leg next to a bf16 headline")
    ap.add_argument("--layers-json", default=None, help="write per-layer device times / rates to this file")
    ap.add_argument("--emit-logits", action="store_true", help="add the gathered per-clip logits of the last step to the line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get("AF_BENCH_FAIL_RANK") == os.environ.get("RANK", ""):        # launcher test hook: this rank dies at start-up
        raise SystemExit(3)

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
    if args.model == "dualrun_rgb":
        return bench_dualrun_rgb(args, rank, world, dev)
    if args.model == "stream":
        return bench_stream(args, rank, world, dev)
    if args.model == "aligner":
        return bench_aligner(args, rank, world, dev)
    if args.model == "conv3x3x3":
        return bench_conv3x3x3(args, rank, world, dev)
    if args.model == "slowfast":
        from af_mi355x.arch import slowfast_r50_spec
        from af_mi355x.classifier import SlowFast8x8
        sd = synth.synthetic_state_dict(slowfast_r50_spec(), seed=0)

        def make_classifier(dtype):
            c = SlowFast8x8(precision=dtype)
            c.load_state_dict(sd)
            return c.to(dev).eval()
        clf = make_classifier(args.dtype)
        net = clf
    elif args.model == "ftcn_tt":
        from af_mi355x.arch import ftcn_tt_spec
        from af_mi355x.classifier import FtcnTTClassifier
        sd = synth.synthetic_state_dict(ftcn_tt_spec(), seed=0)

        def make_classifier(dtype):
            c = FtcnTTClassifier(precision=dtype)
            c.network.load_state_dict(sd)
            return c.to(dev).eval()
        clf = make_classifier(args.dtype)
        net = clf.network
    else:
        sd = synth.synthetic_state_dict(seed=0)

        def make_classifier(dtype, streams=None):
            c = Classifier(precision=dtype, streams=streams)
            c.network.load_state_dict(sd)
            return c.to(dev).eval()
        clf = make_classifier(args.dtype, 1)             # the headline is the single-stream engine (what the roofline measures)
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

    if world > 1:
        line["rccl_ranks"] = world
        line["collective"] = ("gloo (AF_BENCH_REHEARSAL: all ranks on one GPU)" if rehearsal else
                              "RCCL all_gather_into_tensor of (%d,1) fp32 logits per rank per step" % B)
        line["note"] = "roofline / cpu_baseline / parity legs are measured on the N=1 run only (rank 0, world 1)"
    if args.emit_logits:
        line["gathered_logits"] = [float(v) for v in out.float().flatten().cpu()]
    if rank == 0 and world == 1:
        eng = net._engines[(args.dtype, B, (32, 224, 224))]
        total_macs = sum(eng.op_macs)
        line["model_tflops_per_s"] = round(2 * total_macs / B * clips_per_s / 1e12, 2)
        if not args.no_roofline:
            roofline_report(eng, args, line)
        if args.model == "i3d" and not args.no_roofline and args.dtype != "f32":
            # BASELINE.json's literal second metric, timed by whoever runs this line: SYNTHETIC (the reference model has no 3x3x3
            # conv: its bottleneck is 3x1x1 + 1x3x3) - the frame-resident halo kernel with kT = 3 on SURVEY 8d shapes #4 / #18
            shp = conv3x3x3_shapes(args, rank, dev, steps=max(args.steps, 10))
            line["conv3x3x3"] = {"metric": "MFMA % on 3x3x3 Conv3d", "synthetic": True,
                                 "note": "not a layer of the reference model (SURVEY fact 3); Conv3d 3x3x3 + BN + ReLU, batch %d, "
                                         "one launch per shape, events on the launch stream" % B,
                                 "shapes": shp}
            # the reference callers' batch sizes (test/af_realtime.py:318-360 runs 1-3 windows per call, new_demo_test/run_meta.json
            # 8): forward latency of the same engine family, device-resident fp32 input, wall clock over back-to-back steps
            lat = {}
            with torch.inference_mode():
                for b in (1, 2, 4, 8):
                    if b >= B:
                        continue
                    xb = x[:b]
                    for _ in range(3):
                        clf(xb)
                    torch.cuda.synchronize(dev)
                    best, nrep = None, 20                       # best of three rounds: one host hiccup in a 30-step wall-clock loop once read 2.0 ms at B = 2 (1.27 re-measured)
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(nrep):
                            clf(xb)
                        torch.cuda.synchronize(dev)
                        dtb = (time.perf_counter() - t0) / nrep
                        best = dtb if best is None else min(best, dtb)
                    lat[str(b)] = round(1e3 * best, 4)
            line["latency_ms_by_batch"] = lat
        if args.cpu_clips > 0:
            n = min(args.cpu_clips, B)
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            ref, cb = cpu_baseline(sd, u8[:n], max(1, min(cores, 16)), args.model)     # the box's CPU share for one GPU is 16
            line["cpu_baseline"] = cb
            err = float((out[:n].float().cpu() - ref).abs().max())
            line["max_abs_logit_err_vs_cpu_fp32"] = err
            line["logit_tolerance"] = 1e-3
            line["meets_logit_tolerance"] = bool(err <= 1e-3)
            if args.dtype == "bf16":
                line["logit_tolerance_note"] = ("bf16 is BASELINE config[1]'s dtype: its 8-bit weight rounding costs ~0.5 % of the logit "
                                                "(DESIGN.md 5; the reference itself under CPU bf16 autocast is 4.4e-3 off), so this line "
                                                "is a speed number with its error printed; the mode that meets 1e-3 is parity_mode (f16)")
            if args.model == "i3d" and not args.no_parity_mode:
                # opt-in execution mode of the same forward: the batch as two half-batches on two HIP streams (two engines)
                clf2 = make_classifier(args.dtype, 2)
                with torch.inference_mode():
                    for _ in range(args.warmup):
                        y2 = clf2(x)["final_output"]
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        y2 = clf2(x)["final_output"]
                    torch.cuda.synchronize(dev)
                    dt2 = time.perf_counter() - t0
                line["two_stream_mode"] = {"streams": 2, "value": round(B * args.steps / dt2, 2), "unit": "clips/s",
                                           "ms_per_step": round(1e3 * dt2 / args.steps, 4),
                                           "max_abs_logit_err_vs_cpu_fp32": float((y2[:n].float().cpu() - ref).abs().max()),
                                           "note": "Classifier(streams=2) / AF_MI355X_STREAMS=2: off by default"}
                del clf2
            if args.dtype == "bf16" and not args.no_parity_mode:
                # bf16 (8 significant bits) does not reliably meet the 1e-3 logit tolerance; fp16 - the reference's own
                # deployment precision (torch.amp.autocast, test/af_realtime.py:70,84) - does, at the same speed: the
                # same workload timed the same way in that mode, next to the headline
                clf16 = make_classifier("f16")
                with torch.inference_mode():
                    for _ in range(args.warmup):
                        y16 = clf16(x)["final_output"]
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        y16 = clf16(x)["final_output"]
                    torch.cuda.synchronize(dev)
                    dt16 = time.perf_counter() - t0
                e16 = float((y16[:n].float().cpu() - ref).abs().max())
                line["parity_mode"] = {"dtype": "f16", "value": round(B * args.steps / dt16, 2), "unit": "clips/s",
                                       "ms_per_step": round(1e3 * dt16 / args.steps, 4),
                                       "max_abs_logit_err_vs_cpu_fp32": e16, "logit_tolerance": 1e-3,
                                       "meets_logit_tolerance": bool(e16 <= 1e-3)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
