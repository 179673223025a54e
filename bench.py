#!/usr/bin/env python3
"""Benchmark of the denoiser hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W        # starts the N ranks itself (torchrun children)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Step = one training step of BASELINE.json configs[1] per GPU: batch 32 x 8192-frame synthetic
latents, bf16 compute — noise draw, xt, denoiser forward, distance-marching loss, full backward,
(N>1: RCCL all-reduce of the 46.9 M fp32 gradients, overlapped), global-norm clip, AdamW, EMA.
Weak scaling: every rank runs its own batch of 32; `value` = rank-steps per second over the job.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def default_model_args():
    from osu_dreamer_amd.model import BackboneArgs, DiffusionModelArgs
    # osu_dreamer/models/diffusion/model.yml:62-91
    return dict(emb_dim=6, a_dim=128, style_dim=32,
                diffusion_args=DiffusionModelArgs(global_cond_dim=512, backbone_dim=512, u_head_dim=64,
                                                  backbone_args=BackboneArgs(head_dim=64, n_heads=16, depth=8, expand=4, radius=2)))


def make_trainer(device, seed):
    from osu_dreamer_amd.lr_schedule import LRScheduleArgs
    from osu_dreamer_amd.train import DiffusionTrainer
    torch.manual_seed(seed)
    tr = DiffusionTrainer(val_batches=8, opt_args=dict(lr=3e-4, weight_decay=0.01),
                          schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=1000, decay_start=30000),
                          osl_weight=1., del_weight=30., **default_model_args())
    # zero-initialised tensors (ssg/proj_out/u_mod) would make every layer the identity and all
    # gradients trivially sparse: give them N(0, 0.02) so the step does representative work.
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in tr.diffusion.named_parameters():
            if any(z in n for z in ("ssg1.", "ssg2.", "proj_out.", "u_mod.")) or n == "u_out.weight":
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    tr.diffusion_ema.module.load_state_dict(tr.diffusion.state_dict())
    tr.gradient_clip_val = 1.0
    return tr.to(device)


def synthetic_batch(B, L, device, seed):
    """h ~ N(0,1); z per-frame RMS-normalised N(0,1); s RMS-normalised N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(B, 128, L, generator=g)
    z = torch.randn(B, 6, L, generator=g)
    z = z * torch.rsqrt((z * z).mean(1, keepdim=True) + 1e-6)
    s = torch.randn(B, 32, generator=g)
    s = s * torch.rsqrt((s * s).mean(1, keepdim=True) + 1e-6)
    labels = torch.rand(B, 5, generator=g) * 10
    return tuple(t.to(device) for t in (h, z, s, labels))


def flops_forward(frames, L):
    """Algorithmic forward FLOPs (SURVEY.md §8d closed form, verified against FlopCounterMode)."""
    return frames * (68_244_644 + 32_768 * L)


def time_kernel(fn, iters=3):
    """Average duration (s) of `fn` on torch's current stream, measured with HIP events."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


BWD_PASSES_ALGORITHMIC = 5     # S = QK^T recompute, dP = dO V^T, dV = P^T dO, dK = dS^T Q, dQ = dS K  (backward = 2x forward + recompute)


def bwd_passes_executed(eng=None):
    """MFMA passes the attention backward of the plan actually issues (5 for the single-kernel backward od_flash_attn_bwd_fused;
    7 when dK/dV and dQ are separate kernels that each recompute S and dP)."""
    from osu_dreamer_amd import _lib
    if eng is not None:
        return eng.attn_bwd_passes()
    return int(_lib.lib().cdll.od_flash_attn_bwd_passes())


def _lib_sha():
    from osu_dreamer_amd import _lib
    return _lib.source_sha()


def load_pmc(B, L):
    """PMC results of the kernels THIS library was built from: the newest profiles/r*_traffic.json (written by tools/pmc_step.sh +
    tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES rocprofv3 passes over one bench step: counters
    cannot be read in-process) whose `kernel_src_sha` equals od_build_source_sha() of the loaded library.  A record taken on another
    build is NOT quoted.  Returns (HBM bytes per od_flash_attn_bwd launch, per-kernel-class table, source tag) or (None, None, tag)."""
    import glob
    sha = _lib_sha()
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")), reverse=True):
        try:
            tr_ = json.load(open(path))
        except Exception:
            continue
        if tr_.get("kernel_src_sha") == sha and (B, L) == (tr_.get("B"), tr_.get("L")):
            e = tr_["od_flash_attn_bwd"]
            return (int(e["read_bytes"] + e["write_bytes"]), tr_.get("classes"),
                    {"source": os.path.relpath(path, REPO), "kernel_src_sha": sha})
    return None, None, {"source": None, "kernel_src_sha": sha,
                        "note": "no profiles/r*_traffic.json was taken on this build (or at this batch x frames): counters not quoted"}


def load_power_context():
    """The dense bf16 rate this board SUSTAINS under its power cap: bare v_mfma_f32_16x16x32_bf16 on pseudo-random operands, one wave per
    SIMD, nothing else in the loop (tools/ubench/mfma_power.hip; committed as profiles/r*_mfma_power.txt, measured on some box of the
    pool - boxes differ by +-5 %).  Context for `frac`, whose `peak` stays the guide's 2500."""
    import glob, re
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_mfma_power.txt")), reverse=True):
        try:
            m = re.search(r"shape 16:.*?:\s*(\d+) TF/s", open(path).read())
        except Exception:
            continue
        if m:
            return {"bare_mfma_16x16x32_sustained_tflops": int(m.group(1)), "source": os.path.relpath(path, REPO),
                    "note": "power-throttled (1400 W cap); not measured in this run"}
    return None


def load_sampler_parity():
    """tests/test_sampler50.py's record of the 50-step sampler against the REFERENCE's own run (tests/golden/sample50_full_d8_b4_l1115.npz),
    committed as profiles/r*_sampler_parity.json; quoted only when it was measured on this build."""
    import glob
    sha = _lib_sha()
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_sampler_parity.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("kernel_src_sha") == sha:
            return rec, os.path.relpath(path, REPO)
    return None, None


class PowerSampler:
    """Samples the GPU's hwmon files (board power, shader clock) from a host thread while a region runs, so that the line says what box / what
    throttle state the number was taken in.  Plain reads of /sys/class/drm/card*/device/hwmon/*/{power1_input,freq1_input,power1_cap} — no child
    process (a `rocm-smi` child is an exec from a GPU-initialised process: refused under rocprofv3) and nothing that changes a GPU setting.
    Every field is None where the files are absent or unreadable."""

    def __init__(self, device_index=0, period_s=0.01):
        import threading
        self.period = period_s                      # 100 Hz: the power trace is integrated into joules when the board has no energy counter
        self.dir = self._hwmon_of(device_index)
        self.samples, self._stop, self._thr = [], threading.Event(), None
        self.times, self.t_enter, self.t_exit, self.e_enter, self.e_exit = [], None, None, None, None

    @staticmethod
    def _hwmon_of(device_index):
        import glob
        cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        cands = [os.path.dirname(c) for c in cands]
        if not cands:
            return None
        if len(cands) == 1:
            return cands[0]
        try:                                   # several GPUs visible: match the PCI address torch reports for this device
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for c in cands:
                if want in os.path.realpath(os.path.join(c, "..", "..")):
                    return c
        except Exception:
            pass
        return cands[device_index] if device_index < len(cands) else None

    def _num(self, name):
        try:
            return float(open(os.path.join(self.dir, name)).read().strip())
        except Exception:
            return None

    def _read(self):
        if self.dir is None:
            return None
        p, c = self._num("power1_input"), self._num("freq1_input")
        return (p / 1e6 if p is not None else None, c / 1e6 if c is not None else None)

    def __enter__(self):
        import threading

        def loop():
            while not self._stop.is_set():
                r = self._read()
                if r is not None:
                    self.samples.append(r)
                    self.times.append(time.perf_counter())
                self._stop.wait(self.period)
        self.e_enter = self._num("energy1_input") if self.dir is not None else None      # microjoules since boot, where the driver exposes it
        self.t_enter = time.perf_counter()
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        self.t_exit = time.perf_counter()
        self.e_exit = self._num("energy1_input") if self.dir is not None else None
        self._stop.set()
        self._thr.join(timeout=10)

    def joules(self):
        """Energy the board drew between __enter__ and __exit__: the hwmon energy counter's delta where there is one, otherwise the power trace
        integrated over the region (trapezoid rule over the 100 Hz samples, the first / last reading held to the region's ends).  (value, source)"""
        if self.e_enter is not None and self.e_exit is not None and self.e_exit > self.e_enter:
            return (self.e_exit - self.e_enter) / 1e6, "hwmon energy1_input delta"
        pts = [(t, p) for t, (p, _) in zip(self.times, self.samples) if p is not None and self.t_enter is not None and self.t_exit is not None]
        if len(pts) < 2:
            return None, "no power readings"
        pts = [(self.t_enter, pts[0][1])] + [(t, p) for t, p in pts if self.t_enter < t < self.t_exit] + [(self.t_exit, pts[-1][1])]
        e = sum(0.5 * (p0 + p1) * (t1 - t0) for (t0, p0), (t1, p1) in zip(pts, pts[1:]))
        return e, f"integral of hwmon power1_input over the region ({len(pts) - 2} samples at {1.0 / self.period:.0f} Hz, trapezoid)"

    def cap_w(self):
        v = self._num("power1_cap") if self.dir is not None else None
        return v / 1e6 if v is not None else None

    def summary(self):
        pw = [p for p, _ in self.samples if p is not None]
        ck = [c for _, c in self.samples if c is not None]
        return {"power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "clk_mhz_mean": round(sum(ck) / len(ck), 1) if ck else None,
                "power_cap_w": self.cap_w(), "samples": len(self.samples),
                "source": "hwmon power1_input / freq1_input of the GPU, sampled from a host thread over the timed region (instantaneous readings, not cycle averages)"}


def mfma_calibration(device, seconds=2.0):
    """~2 s of bare v_mfma_f32_16x16x32_bf16 on pseudo-random operands through the C ABI (od_mfma_calibrate): the dense bf16 rate THIS box
    sustains under its power cap, measured in this run.  ms_per_step x this figure is comparable between boxes and rounds."""
    import ctypes
    from osu_dreamer_amd import _lib
    scratch = torch.empty(256 * 1024, dtype=torch.float32, device=device)
    flops = ctypes.c_double()
    st = torch.cuda.current_stream(device).cuda_stream

    def launch(iters):
        _lib.lib().od_mfma_calibrate(scratch.data_ptr(), iters, ctypes.byref(flops), st)
    launch(2000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    with PowerSampler(device.index or 0) as ps:
        e0.record()
        while time.time() - t0 < seconds:
            for _ in range(5):
                launch(20000)
            n += 5
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    out = {"sustained_tflops": round(n * flops.value / ms / 1e9, 1), "seconds": round(ms / 1e3, 2),
           "kernel": "od_mfma_calibrate: bare v_mfma_f32_16x16x32_bf16, pseudo-random operands, one wave per SIMD on every CU"}
    out.update({k: v for k, v in ps.summary().items() if k in ("power_w_mean", "clk_mhz_mean")})
    return out


def roofline_of_dominant_kernel(tr, B, L):
    """Re-launch the step's dominant kernel (od_flash_attn_bwd: the attention backward of one layer) alone on the step's
    own buffers and time it with HIP events on the launch stream.  `achieved` counts ALGORITHMIC FLOPs: 5 MFMA passes of
    2*B*H*L^2*hd each (flash attention's backward with the score recompute); the passes actually executed are reported
    separately."""
    from osu_dreamer_amd import ops
    eng = tr.diffusion.engine
    t = eng.ws.t
    H, hd, dh = eng.H, eng.hd, eng.dh
    i = eng.depth - 1
    qk, qkv, y, lse = t[f"qk16.{i}" if eng.attn_f16 else f"qk.{i}"], t[f"qkv.{i}"], t[f"y.{i}"], t[f"lse.{i}"]
    dy, dqkv, delta = t["d.y"], t["d.qkv"], t["d.delta"]
    scale = 1 / math.sqrt(hd)

    fused = eng.fused_attn_bwd()

    def bwd():       # exactly as the step launches it
        eng.attn_bwd_launch(i, dy, delta, dqkv, core_only=True)

    def fwd():
        ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], eng._v_of(qkv), y, lse, B, H, L, hd, scale, q_prescaled=True)
    unit = 2.0 * B * H * L * L * hd                 # one L x L x hd MFMA pass over all heads
    t_bwd, t_fwd = time_kernel(bwd), time_kernel(fwd)
    executed = bwd_passes_executed(eng)
    traffic, classes, traffic_src = load_pmc(B, L)
    ach_bwd = BWD_PASSES_ALGORITHMIC * unit / t_bwd / 1e12
    ach_fwd = 2 * unit / t_fwd / 1e12
    return {
        "bound": "mfma", "kernel": ("od_flash_attn_bwd_fused (attention backward of one layer: start values, then ONE persistent kernel — S, dP, dV, dK, dQ; dQ summed over key blocks by the L2 chain)" if fused else
                                   "od_flash_attn_bwd (attention backward of one layer: delta, then dK/dV and dQ on two streams)"),
        "achieved": round(ach_bwd, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(ach_bwd / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
        "traffic_note": ("fabric-side bytes of the call (L2 misses + write-through stores).  The operands are ~4.3 GB (q, k, v, o, dO read once; dq, dk, dv written once); "
                         "the rest is the dQ chain: a running tile of 12 KB per query tile (16 x 22-bit values + the write number in three 16-byte pieces per lane; "
                         "round 6: 16 KB of fp32 before) handed key block -> key block through the XCD's L2 with non-temporal loads, "
                         f"2 x 12 KB x {B * H * ((L + 63) // 64) * (((L + 191) // 192) - 1) / 1e6:.2f} M hand-offs, most of which stay in the L2 (DESIGN.md sections 3 and 9)"
                         if fused else None),
        "ms_per_launch": round(t_bwd * 1e3, 3),
        "flops_counted": f"algorithmic: {BWD_PASSES_ALGORITHMIC} passes x 2*B*H*L^2*hd",
        "mfma_passes_executed": executed,
        "achieved_executed": round(executed * unit / t_bwd / 1e12, 1),
        "also": {"od_flash_attn_fwd": {"achieved": round(ach_fwd, 1), "frac": round(ach_fwd / PEAK_BF16_TFLOPS, 4),
                                       "ms_per_launch": round(t_fwd * 1e3, 3)},
                 # per kernel class over one step, from the same PMC record as `traffic` (null when that record is not of this build):
                 # HBM fraction of 8 TB/s for the memory-bound classes, MFMA-pipe busy fraction for the matrix classes
                 "pmc_classes": classes,
                 "power_limit": load_power_context()},
    }


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(L_cpu=2048, B=32, L=8192):
    """The CPU oracle (a port of the reference path, pinned by tests/golden) timed on this box's host cores on a bounded sample: ONE fp32
    training step at batch 1 x L_cpu frames.  `value` scales the sample to the bench step (B x L frames) by ALGORITHMIC FLOPs — the closed
    form of SURVEY.md section 8d, 3 x frames x (68,244,644 + 32,768 L): attention's cost per frame grows with L, so scaling by frames alone
    (reported beside it) over-states the CPU rate.  `--cpu-frames 8192` times batch 1 at the bench's own L instead (minutes, tens of GB of
    host memory for the oracle's dense attention), and then only the batch factor is extrapolated."""
    from oracle import denoiser_oracle as O
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    d = O.FULL
    P = O.init_params(d, seed=7)
    data = O.synthetic_batch(d, 1, L_cpu, seed=8)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    vv = {k: torch.zeros_like(v) for k, v in P.items()}
    ema = {k: v.clone() for k, v in P.items()}
    t0 = time.time()
    _, _, grads = O.loss_and_grads(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])
    _, coef = O.clip_coef(grads, 1.0)
    O.adamw_ema_step(P, grads, m, vv, ema, 1, 3e-4, clip=coef, first_ema=True)
    dt = time.time() - t0
    by_flops = dt * flops_forward(B * L, L) / flops_forward(L_cpu, L_cpu)      # seconds per bench step, scaled by algorithmic FLOPs
    by_frames = dt * (B * L) / L_cpu
    return {"value": 1.0 / by_flops,
            "unit": f"train-steps/s ({B}x{L}-frame step equivalent, EXTRAPOLATED from the timed sample by algorithmic FLOPs: frames x (68,244,644 + 32,768 L))",
            "cores": cores, "cpu": cpu_model(), "kind": "port",
            "sample": f"1 fp32 train step, batch 1 x {L_cpu} frames, {dt:.1f} s, {L_cpu / dt:.0f} frames/s at that length",
            "value_scaled_by_frames_only": 1.0 / by_frames,
            "extrapolation": {"sample_frames": L_cpu, "sample_seconds": round(dt, 2), "flops_factor": round(flops_forward(B * L, L) / flops_forward(L_cpu, L_cpu), 2),
                              "frames_factor": round(B * L / L_cpu, 2), "same_length_as_bench": L_cpu == L}}


def allreduce_alone(reducer, device, iters=3):
    """The gradient exchange with nothing beside it: the arena's segments all-reduced back to back on the launch stream (every rank calls
    this), HIP events around them.  The figure the N-rank line's exposed time and SURVEY section 8(e)'s 0.3 - 2.1 ms are read against.
    (At world size 1 an in-place ncclAllReduce launches nothing: the time is the enqueue cost.)"""
    g = reducer.model.arena.ensure_grad()
    segs = [g[s:e] for s, e in reducer.segments.values()]
    def run():
        for t in segs:
            reducer.comm.allreduce_mean_(t)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, sum(t.numel() * 4 for t in segs), len(segs)


def sampler_bench(device):
    """BASELINE configs[3]: 50-step sampler, 4 diffs in parallel on a 3-min song (L=1115 latent frames),
    hipGraph-captured step.  fp32 = exact fp32 MFMA chain, fp32_bf16x3 = fp32 tensors with 3 bf16 MFMAs per
    product (both inside the 1e-4 sampler parity bound), bf16 = autocast-equivalent compute."""
    from osu_dreamer_amd.model import DiffusionModel
    a = default_model_args()
    torch.manual_seed(5)
    m = DiffusionModel(a["emb_dim"], a["a_dim"], a["style_dim"], a["diffusion_args"])
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if any(z in n for z in ("ssg1.", "ssg2.", "proj_out.", "u_mod.")) or n == "u_out.weight":
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    m = m.to(device)
    B, L = 4, 1115
    h = torch.randn(1, 128, L, generator=g).to(device)
    s = torch.randn(B, 32, generator=g).to(device)
    out = {}
    for name, dt, mm in (("fp32", torch.float32, "f32"), ("fp32_bf16x3", torch.float32, "bf16x3"),
                         ("bf16", torch.bfloat16, "f32")):
        m.compute_dtype, m.f32_matmul = dt, mm
        m.sample(h, s, 50)
        torch.cuda.synchronize()
        t0 = time.time()
        m.sample(h, s, 50)
        torch.cuda.synchronize()
        dt_s = time.time() - t0
        out[name] = {"latents_per_s": round(B * L / dt_s, 1), "ms_per_sample_call": round(dt_s * 1e3, 2),
                     "tflops": round(51 * flops_forward(B * L, L) / dt_s / 1e12, 1)}
    # which of these modes meets north_star's "within 1e-4 relative L2 of reference" — measured by tests/test_sampler50.py against the
    # reference's own 50-step run at this very size; quoted only when that record is of the build being benchmarked
    rec, src = load_sampler_parity()
    for name in out:
        mode = (rec or {}).get("modes", {}).get(name)
        out[name]["meets_1e-4"] = None if mode is None else bool(mode["meets_1e-4"])
        out[name]["rel_l2_vs_reference"] = None if mode is None else float(mode["rel_l2_vs_reference_fp32"])
    out["parity_source"] = src if rec else "no profiles/r*_sampler_parity.json of this build (run pytest -m gpu tests/test_sampler50.py)"
    return {"workload": "50-step sampler, B=4, L=1115, audio batch 1 (broadcast), hipGraph", **out}


def ldm_bench(device):
    """Whole inference pipeline, LDM.sample (inference/model.py:34-51) on BASELINE configs[3]'s song: a 3-minute
    spectrogram (72 x 30093 frames), 4 difficulties, 50 denoiser steps, default model.yml widths.  Stage times from
    HIP events around each stage on the launch stream."""
    from osu_dreamer_amd.ldm import LDM
    a = default_model_args()
    torch.manual_seed(7)
    m = LDM(dict(emb_dim=6, style_dim=32, n_downs=3, stride=3,
                 latent_args=dict(h_dim=128, ae_args=dict(n_layers=8, expand=4, radius=2), style_head_dim=64, style_heads=16),
                 style_args=dict(label_features=128, h_dim=256, depth=8, expand=4),     # models/style/model.yml
                 diffusion_args=a["diffusion_args"]))
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():      # tensors the reference zero-initialises: give them small values so no stage is an identity
        for n, p in m.named_parameters():
            if float(p.abs().max()) == 0.0:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    m = m.to(device).eval()
    B, L = 4, 30093
    audio = torch.randn(72, L, generator=g).to(device)
    labels = (torch.rand(B, 5, generator=g) * 10).to(device)
    out = {}
    for name, dt, mm in (("fp32", None, "f32"), ("fp32_bf16x3", None, "bf16x3"), ("bf16", torch.bfloat16, "f32")):
        m.set_precision(dt, mm)
        m.sample(audio, labels, 50)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        from osu_dreamer_amd.ldm import pad_to_multiple
        t0 = time.time()
        ev[0].record()
        ap = pad_to_multiple(audio, m.latent.chunk_size)
        skips, h = m.latent.audio_encoder(ap[None]); ev[1].record()
        s_ = m.style.sample(labels); ev[2].record()
        z = m.diffusion.sample(h, s_, 50); ev[3].record()
        chart, _ = m.latent.decode(z, s_, skips=skips); ev[4].record()
        torch.cuda.synchronize()
        wall = time.time() - t0
        st = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
        out[name] = {"ms_total": round(wall * 1e3, 2), "charts_per_s": round(B / wall, 2),
                     "spec_frames_per_s": round(B * L / wall, 1),
                     "ms_audio_encoder": round(st[0], 2), "ms_style_sampler": round(st[1], 2),
                     "ms_denoiser_sampler": round(st[2], 2), "ms_decode": round(st[3], 2)}
    return {"workload": "LDM.sample: 72x30093 spectrogram (3 min), 4 difficulties, 50 steps, default widths", **out}


def h2d_leg(batch, device, ms_step):
    """What the boundary costs when the batch arrives in HOST memory (the feeder's pinned tensors, osu_dreamer_amd/data.py; `value` is
    quoted with inputs resident in HBM): one batch copied host -> device on the launch stream, timed with HIP events, and the step
    rate with that copy serialised in front of every step (Trainer.fit does exactly that; a prefetching feeder would hide it)."""
    host = [t.detach().cpu().pin_memory() for t in batch]
    for t in host:
        t.to(device, non_blocking=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for t in host:
            t.to(device, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    nbytes = sum(t.numel() * t.element_size() for t in host)
    return {"batch_mb": round(nbytes / 1e6, 1), "h2d_ms": round(ms, 3), "h2d_gb_per_s": round(nbytes / ms / 1e6, 1),
            "steps_per_s_with_serial_h2d": round(1e3 / (ms_step + ms), 4)}


def forward_target_shape(tr, device, B=64, L=8192):
    """north_star's target shape: denoiser FORWARD at batch 64 x 8192 frames, bf16.  Reports the binding
    (MFMA) fraction and, because the target was phrased against HBM, the HBM fraction of the algorithmic
    bytes as well (87,314 elements/frame)."""
    h, z, s, _ = synthetic_batch(B, L, device, seed=99)
    m = tr.diffusion
    with torch.no_grad():
        m(h, s, z)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(2):
            m(h, s, z)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / 2
    fl = flops_forward(B * L, L)
    by = 87_314 * 2.0 * B * L
    return {"workload": f"denoiser forward, batch {B} x {L} frames, bf16", "ms": round(dt * 1e3, 1),
            "tflops": round(fl / dt / 1e12, 1), "mfma_frac": round(fl / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
            "algorithmic_gb": round(by / 1e9, 1), "hbm_frac_of_algorithmic_bytes": round(by / dt / 1e9 / PEAK_HBM_GBS, 4),
            "frames_per_s": round(B * L / dt, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--attn-dtype", default="bf16", choices=["bf16", "f16"],
                    help="f16: the attention core on IEEE-half operands (BASELINE configs[4] 'attention in fp16'; bf16 compute around it)")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / sampler legs")
    ap.add_argument("--cpu-frames", type=int, default=2048, help="length of the CPU baseline's batch-1 sample (8192 = the bench's own L: minutes)")
    args = ap.parse_args()

    # --gpus N: one process per GPU.  Outside a torchrun job this process only starts the N ranks (as children,
    # before anything here touches the GPU) and hands back their exit code.
    from osu_dreamer_amd import launch
    rc = launch.spawn_ranks_if_needed(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    world = launch.check_world(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from osu_dreamer_amd import _lib
    _lib.lib()                                   # fail loudly if the HIP library is missing
    reducer = None
    ddp = world > 1 or os.environ.get("OD_FORCE_DDP") == "1"      # OD_FORCE_DDP: exercise the RCCL path on one GPU
    if ddp:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)
        assert dist.get_world_size() == world == args.gpus or os.environ.get("OD_FORCE_DDP") == "1"
    B, L = args.batch, args.frames
    tr = make_trainer(device, seed=1234)
    tr.diffusion.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.attn_dtype == "f16":
        tr.diffusion.attn_dtype = torch.float16
    cfg = tr.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    if ddp:
        from osu_dreamer_amd.ddp import GradBucketReducer
        reducer = GradBucketReducer(tr.diffusion)          # RCCL communicator through the C ABI (od_comm_*)
        reducer.broadcast_state(opt, tr.diffusion_ema, src=0)
    batch = synthetic_batch(B, L, device, seed=1234 + rank)

    def step(i):
        opt.zero_grad()
        loss = tr.training_step(batch, i)
        loss.backward()
        opt.step()
        sched.step()
        tr.on_train_batch_end()
        return loss

    def barrier():
        if ddp:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = step(i)
    if reducer is not None:
        reducer.exposed_ms()                   # drop the warm-up's events
        reducer.time_exposed = True            # HIP events either side of the compute stream's wait for the exchange (no host sync)
    sampler = PowerSampler(device.index or 0)
    barrier()
    with sampler:
        t0 = time.time()
        for i in range(args.steps):
            loss = step(args.warmup + i)
        barrier()
        dt = time.time() - t0
    per_rank = None
    if ddp:
        import torch.distributed as dist
        # every rank's own wall time and exposed-communication time, so that a slow rank or an un-hidden exchange shows in the line
        exposed = reducer.exposed_ms()
        mine = torch.tensor([dt / args.steps * 1e3, sum(exposed) / max(len(exposed), 1), max(exposed, default=0.0)],
                            device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu()
        dt = float(per_rank[:, 0].max()) * args.steps / 1e3
    final_loss = float(loss.detach())
    alone = allreduce_alone(reducer, device) if ddp else None        # every rank: a collective

    if rank == 0:
        ms = dt / args.steps * 1e3
        frames = B * L
        f_fwd = flops_forward(frames, L)
        # executed MFMA work: forward + 2x GEMM backward + attention backward as its executed passes (vs 2 forward)
        attn_fwd = frames * 32_768 * L
        executed = f_fwd + 2 * (f_fwd - attn_fwd) + bwd_passes_executed(tr.diffusion.engine) / 2 * attn_fwd
        f16a = args.attn_dtype == "f16"
        named = {(32, 8192): "BASELINE.json configs[1]",
                 (8, 32768): ("BASELINE.json configs[4] shape, attention in fp16 with MFMA: q, k, v, P, dS and the staged dO are IEEE half (v_mfma_f32_*_f16), bf16 compute around the core"
                              if f16a else "BASELINE.json configs[4] shape; attention in bf16 MFMA (--attn-dtype f16 runs it in fp16)"),
                 (2, 4096): "BASELINE.json configs[0] shape, on the GPU"}.get((B, L), "custom --batch/--frames")
        if f16a and (B, L) != (8, 32768):
            named += "; attention core on IEEE-half operands (--attn-dtype f16)"
        if world > 1 and (B, L) == (32, 8192):
            named = f"BASELINE.json configs[2] pattern at {world} ranks (configs[1] per rank)"
        line = {
            "metric": "denoiser train-steps/sec + 50-step sample latents/sec, 1/2/4/8 MI355X",
            "value": round(world * args.steps / dt, 4), "unit": f"train-steps/s (rank-steps of batch {B} x {L} frames)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "attn_dtype": args.attn_dtype, "data": "synthetic",
            "config": {"workload": f"denoiser training step, batch={B} per GPU, {L}-frame synthetic latents, {args.dtype} "
                                   f"({named}); per-rank batch fixed under --gpus N",
                       "per_gpu_batch": B, "global_batch": B * world, "frames": L, "params": 46_877_103,
                       "parallelism": f"dp{world}"},
            "frames_per_s": round(world * frames * args.steps / dt, 1),
            "algorithmic_tflops_per_step": round(3 * f_fwd / 1e12, 1),
            "achieved_tflops_algorithmic": round(3 * f_fwd / (dt / args.steps) / 1e12, 1),
            "achieved_tflops_executed": round(executed / (dt / args.steps) / 1e12, 1),
            "mfma_frac_whole_step": round(3 * f_fwd / (dt / args.steps) / 1e12 / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS), 4),
            "final_loss": round(final_loss, 4),
            "workspace_gb": round(tr.diffusion.engine.ws.bytes / 2**30, 1),
        }
        # what box / what throttle state this line was taken in (rank 0's GPU): board power and shader clock sampled over the timed region
        line.update({k: v for k, v in sampler.summary().items() if k in ("clk_mhz_mean", "power_w_mean", "power_cap_w")})
        line["power_sampling"] = {k: v for k, v in sampler.summary().items() if k in ("samples", "source")}
        # energy: what the board drew over the timed region, per step and per algorithmic PFLOP (rank 0's GPU) — the quantity that sets the
        # rate of a power-capped kernel; per kernel class: tools/energy_classes.py -> profiles/r06_energy_classes.txt
        ej, esrc = sampler.joules()
        line["joules_per_step"] = round(ej / args.steps, 1) if ej is not None else None
        line["joules_per_algorithmic_pflop"] = round(ej / args.steps / (3 * f_fwd / 1e15), 2) if ej is not None else None
        line["energy_source"] = esrc
        if ddp:
            line["collective"] = {"backend": "RCCL via od_allreduce_grads", "version": reducer.comm.version,
                                  "exchange": "od_allreduce_grads per arena segment (187.5 MB fp32 per step), overlapped with backward",
                                  "bytes_per_step": alone[1], "buckets_per_step": alone[2],
                                  # the same ten all-reduces with nothing beside them (HIP events, after the timed region): what the ring costs
                                  # when it does not have to share the chip (SURVEY section 8e expects 0.3 - 2.1 ms at 8 ranks over xGMI)
                                  "allreduce_alone_ms": round(alone[0], 3),
                                  "allreduce_alone_gb_per_s": round(alone[1] / max(alone[0], 1e-6) / 1e6, 1),
                                  "world_size": world, "rccl_ranks_seen": reducer.comm.ranks_seen,
                                  # time the compute stream waited for the side-stream exchange before clip / AdamW (HIP events): mean
                                  # over steps, averaged and maximised over ranks; 0 = fully hidden under the backward
                                  "allreduce_exposed_ms": round(float(per_rank[:, 1].mean()), 3),
                                  "allreduce_exposed_ms_max_rank": round(float(per_rank[:, 1].max()), 3),
                                  "allreduce_exposed_ms_worst_step": round(float(per_rank[:, 2].max()), 3),
                                  "ms_per_step_min_rank": round(float(per_rank[:, 0].min()), 2),
                                  "ms_per_step_max_rank": round(float(per_rank[:, 0].max()), 2),
                                  "ms_per_step_per_rank": [round(float(x), 2) for x in per_rank[:, 0]],
                                  "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                  "comm_stream_priority": int(os.environ.get("OD_COMM_STREAM_PRIORITY", "0"))}
        if not args.no_extras:
            # rank 0 only; at N > 1 the other ranks wait in the barrier below (their GPUs idle): the dominant kernel is re-timed on
            # rank 0's GPU and the CPU baseline on the host cores, so that an N-rank line carries the same two objects as the N = 1 line
            line["calibration"] = mfma_calibration(device)
            line["ms_per_step_x_sustained_tflops"] = round(ms * line["calibration"]["sustained_tflops"], 1)     # comparable between boxes and rounds
            line["roofline"] = roofline_of_dominant_kernel(tr, B, L)
            if world == 1:
                line["pcie_inclusive"] = h2d_leg(batch, device, ms)
                line["forward_64x8192"] = forward_target_shape(tr, device)
                line["sampler"] = sampler_bench(device)
                line["ldm_sample"] = ldm_bench(device)
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames, B, L)
            if world > 1:
                line["cpu_baseline"]["note"] = "timed on rank 0's host while the other ranks waited in a barrier"
    else:
        line = None
    if ddp:
        import torch.distributed as dist
        dist.barrier()
        reducer.close()
        dist.destroy_process_group()      # RCCL prints its banner here: keep the JSON the LAST line
    if line is not None:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
