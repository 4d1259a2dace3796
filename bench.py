#!/usr/bin/env python3
"""Headline benchmark: SpectralMixingLayer fwd+bwd throughput on synthetic (B, N, D) fp32.

    python bench.py --gpus 1 --steps 50 --warmup 10            # BASELINE config C2 (the driver's line)
    python bench.py --config c3 | c5                           # the other measured single-GPU configs
    python bench.py --gpus N --steps K --warmup W              # N > 1 without a launcher: bench.py starts the
                                                               # N rank processes itself (launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # or under a launcher (the driver's way)

A step = {y = layer(x); y.backward(g); zero grads} on one batch resident in HBM -- the semantics of
the reference's harness benchmark_spectral.py:190-210, except that g is random.
Workloads (BASELINE.json configs):
  c2  (B=64, N=4096,  D=256, F=128)  SpectralMixingLayer                     -- default, headline
  c3  (B=8,  N=65536, D=256, F=128)  SpectralMixingLayer, long sequence (residue-split plan)
  c5  (B=64, N=4096,  D=512, F=256)  ifft(WirtingerSpectralFilter(fft(x))).real through
                                     spectral_mix_with_filter (reference wirtinger_ops.py:170-203)
At N > 1 GPUs the workload is per rank (weak scaling, batch sharded) and the filter/bias gradients are
sum-all-reduced over RCCL inside backward; the schedule of that collective ("overlap" or "fused",
tensor-cuda-fft-_amd/distributed.py) is picked by timing both before the timed region.
Rank 0 prints one JSON line; at N = 1 with the default config it also carries the C3 and C5 measurements
("other_configs": same binary, same box, same protocol; "f2" = fft_lm's causal convolution); "ranks_seen" is a SUM all-reduce of 1.0 per rank.
"""
import argparse
import glob
import hashlib
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE_DIR = 8   # one direction: read x + write y   (16 B/sample fwd+bwd)

CONFIGS = {
    "c2": dict(B=64, N=4096, D=256, F=128, api="layer"),
    "c3": dict(B=8, N=65536, D=256, F=128, api="layer"),
    "c5": dict(B=64, N=4096, D=512, F=256, api="wirtinger"),
}


def make_unit(pkg, cfg, dev, seed):
    """The module under test and a callable x -> y."""
    torch.manual_seed(seed)
    D, F = cfg["D"], cfg["F"]
    if cfg["api"] == "wirtinger":
        filt = pkg.WirtingerSpectralFilter(D, F).to(dev)
        with torch.no_grad():
            filt.weight.real.normal_(1.0, 0.5)
            filt.weight.imag.normal_(0.0, 0.5)
        return filt, (lambda x: pkg.spectral_mix_with_filter(x, filt)), \
            (filt.weight.real, filt.weight.imag, None)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(dev)
    with torch.no_grad():
        layer.weight_real.normal_(1.0, 0.5)
        layer.weight_imag.normal_(0.0, 0.5)
        layer.bias.normal_(0.0, 0.1)
    return layer, layer, (layer.weight_real, layer.weight_imag, layer.bias)


def cpu_baseline(B, N, D, F, iters):
    """The oracle's fp32 port of the reference op sequence, timed on this host's cores."""
    from oracle import spectral_oracle as so
    torch.manual_seed(1234)
    x = torch.randn(B, N, D); g = torch.randn(B, N, D)
    wr = 1 + 0.5 * torch.randn(D, F); wi = 0.5 * torch.randn(D, F); b = 0.1 * torch.randn(D)
    so.fwd_bwd_port(x, wr, wi, b, g)                       # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        so.fwd_bwd_port(x, wr, wi, b, g)
    dt = (time.perf_counter() - t0) / iters
    return {"value": B * N * D / dt / 1e9, "unit": "GSamples/s", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": dt * 1e3,
            "sample": f"{iters} fwd+bwd steps of the full (B={B},N={N},D={D}) batch, "
                      f"torch {torch.__version__} CPU, {os.cpu_count()} logical cpus"}


def lib_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def csrc_sha256():
    """Identity of the kernel SOURCES: sha256 over the sorted contents of csrc/*.hip, csrc/*.h, csrc/build.sh and
    include/smx.h.  hipcc derives code-object ids from absolute paths, so a rebuild of the same sources in another
    checkout location gives a different binary hash; the sources (and flags, in build.sh) are what define the
    kernels.  tools/summarize_profile.py stamps the same value into every profile summary."""
    h = hashlib.sha256()
    c = os.path.join(ROOT, "tensor-cuda-fft-_amd", "csrc")
    files = sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.h")) +
                   [os.path.join(c, "build.sh"), os.path.join(ROOT, "include", "smx.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def profile_traffic(cfg_name, sha, kernel_prefixes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/): counters cannot be
    collected from inside this process, so the summary of the SAME library build is quoted -- matched by
    the sha256 of libsmx.so that tools/collect_profile.sh stamps into it -- or nothing."""
    reason = "no profiles/*_summary.json for this config"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{cfg_name}_summary.json")))[::-1]:
        try:
            s = json.load(open(f))
            if s.get("libsmx_sha256") != sha and s.get("csrc_sha256") != csrc_sha256():
                reason = (f"{os.path.relpath(f, ROOT)} was collected with another build of libsmx.so "
                          f"({str(s.get('libsmx_sha256'))[:12]} != {sha[:12]}) from other sources")
                continue
            c = s["counters_per_launch"]
            out = {}
            for pre in kernel_prefixes:
                key = [k for k in c if k.startswith(pre) and "hbm_bytes" in c[k]]
                if key:
                    out[pre] = round(c[key[0]]["hbm_bytes"])
            if out:
                same = "same libsmx.so" if s.get("libsmx_sha256") == sha else "same kernel sources, rebuilt libsmx.so"
                return out, os.path.relpath(f, ROOT), same
        except Exception as e:                                           # noqa: BLE001
            reason = f"{os.path.relpath(f, ROOT)}: {type(e).__name__}: {e}"
    return None, None, reason


def profile_kernel_avg(cfg_name, sha, kernel_prefix):
    """(avg_us, file) of a kernel in the committed rocprofv3 --kernel-trace --stats pass of the SAME library build
    (profiles/*_summary.json, `kernel_stats`), or (None, reason): the line's own HIP-event average stands beside it,
    so the two can be compared without leaving the line."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{cfg_name}_summary.json")))[::-1]:
        try:
            s = json.load(open(f))
            if s.get("libsmx_sha256") != sha and s.get("csrc_sha256") != csrc_sha256():
                continue
            for k in s.get("kernel_stats", []):
                if k["name"].startswith(kernel_prefix):
                    return float(k["avg_us"]), os.path.relpath(f, ROOT)
        except Exception:                                                # noqa: BLE001
            continue
    return None, "no profiles/*_summary.json of this build holds the kernel"


def measure_box(dev):
    """The box's own yardsticks, same process, right after load: (a) a plain torch copy of 256 MiB (268 MB read +
    268 MB written = the bytes of ONE fused launch at C2), median of 20 under HIP events; (b) the shader clock the
    chip sustains, from s_memtime / s_memrealtime in a one-wave kernel enqueued straight behind those copies
    (smx_diag_clock).  A fwd+bwd step moves twice the copy's bytes: step / copy says how far the kernels are from
    what THIS box gives a linear copy, whatever its clocks or its HBM do today."""
    from tensor_cuda_fft_amd import _lib
    a = torch.empty(64 * 4096 * 256, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in ev:
        e0.record()
        b.copy_(a)
        e1.record()
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().smx_diag_clock(clk.data_ptr(), 200000, torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    c = clk.tolist()
    del a, b
    torch.cuda.empty_cache()
    return {"copy_256MiB_us": round(ts[len(ts) // 2], 1), "copy_256MiB_min_us": round(ts[0], 1),
            "copy_kind": "torch.Tensor.copy_, linear, 268 MB read + 268 MB written",
            "clock_GHz": round(c[0] / c[1] * 0.1, 3) if c[1] else None,
            "clock_note": "s_memtime ticks / s_memrealtime (100 MHz) ticks of one wavefront spinning behind the copies"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--seq", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--filters", type=int, default=0)
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: the step is captured once into a hipGraph and replayed")
    ap.add_argument("--steps-per-graph", type=int, default=64,
                    help="graph mode: steps captured per hipGraph; K <= 64 timed steps are ONE replay, so the wall "
                         "clock holds one replay launch and one synchronise (round 3 cut --steps 20 into two "
                         "replays of 10: ~5 us per step of harness inside a 4.6 ms region)")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed back-to-back steps before the timed region, so the clocks have ramped "
                         "(W warm-up steps alone are ~3 ms; the chip needs ~100 ms of load to leave idle clocks)")
    ap.add_argument("--sync-mode", choices=["auto", "overlap", "fused"], default="auto",
                    help="N > 1: schedule of the gradient all-reduce (auto = time both, keep the faster)")
    ap.add_argument("--opts", default="", help='plan knobs for A/B runs, "name=value;name=value" (smx_set_option); '
                                               'the line records them in config.opts')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default config: do not also measure C3 and C5 (\"other_configs\" of the line)")
    ap.add_argument("--no-supervise", action="store_true",
                    help="N > 1: run in this process instead of a supervised child (see supervise())")
    return ap.parse_args(argv)


EXIT_NO_DEVICE = 2          # a rank's pre-flight found no GPU for its LOCAL_RANK: nothing to retry
# One attempt (graph mode, or the eager repeat) may take this long; two of them fit the driver's 600 s limit
ATTEMPT_S = float(os.environ.get("SMX_BENCH_ATTEMPT_S", "240"))


class Runtime:
    """What every measurement of this process shares: rank, device, process group."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # Rehearsal knobs (never set by the driver): SMX_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
        # SMX_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N > 1 control flow can be exercised on a one-GPU
        # box; SMX_BENCH_DRY_RUN=1 replaces the GPU step by nothing at all (launcher / rendezvous / JSON
        # plumbing on a machine without a GPU: the line says "dry_run": true and carries no throughput).
        self.dry = os.environ.get("SMX_BENCH_DRY_RUN") == "1"
        self.one_dev = os.environ.get("SMX_BENCH_ONE_DEVICE") == "1"
        self.backend = os.environ.get("SMX_BENCH_BACKEND", "nccl")
        if self.dry:
            self.dev = torch.device("cpu")
            if os.environ.get("SMX_BENCH_TEST_SLEEP_RANK") == str(self.rank):      # tests/test_bench_launch.py:
                time.sleep(3600)                                                     # a rank that never arrives
        else:
            # pre-flight, before anything touches the GPU or a collective (device_count() does not initialise it):
            # a rank without a device says so in one line and exits 2 -- the launcher does not retry that
            idx = 0 if self.one_dev else self.local
            have = torch.cuda.device_count()
            if have <= idx:
                print(f"[bench] rank {self.rank}: LOCAL_RANK {self.local} needs cuda:{idx} but this process sees "
                      f"{have} GPU(s) (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, "
                      f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')!r}): --gpus {args.gpus} "
                      f"cannot run here", file=sys.stderr, flush=True)
                os._exit(EXIT_NO_DEVICE)
            self.dev = torch.device("cuda", idx)
            torch.cuda.set_device(self.dev)
        self.use_dist = self.world > 1 or os.environ.get("SMX_FORCE_SYNC") == "1"
        self.ranks_seen = 1
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            # what the collective library itself saw: a SUM all-reduce of one 1.0 per rank, on the device the
            # gradients live on (RCCL over xGMI for backend nccl)
            ones = torch.ones(1, device=self.dev, dtype=torch.float32)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            self.ranks_seen = int(round(ones.item()))

    def device_sync(self):
        if not self.dry:
            torch.cuda.synchronize(self.dev)

    def sync_all(self):
        self.device_sync()
        if self.world > 1:
            dist.barrier()
            self.device_sync()

    def max_over_ranks(self, v: float) -> float:
        if self.world > 1:
            t = torch.tensor([v], device=self.dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            v = t.item()
        return v

    def gather_over_ranks(self, v: float) -> list:
        """every rank's value, in rank order (one all-gather on the collective backend)"""
        if self.world == 1:
            return [v]
        t = torch.tensor([v], device=self.dev, dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.use_dist:
            dist.destroy_process_group()


def measure_f2(rt, steps):
    """The SURVEY 8(f) row the reference actually trains with: fft_lm's causal FFT convolution
    (FixedSpectralBlock's hot line, reference fft_lm/train_fixed_full.py:507-555) at its default size
    (64, 1024, 512), 128 taps, n_fft 2048 -- forward + backward of tensor_cuda_fft_amd.causal_spectral_conv,
    hipGraph of one step, EXACTLY `steps` replays between device syncs.  16 B/sample algorithmic (fwd + bwd)."""
    import tensor_cuda_fft_amd as pkg
    dev = rt.dev
    B, T, C, K = 64, 1024, 512, 128
    gen = torch.Generator(device=dev).manual_seed(4321)
    x = torch.randn(B, T, C, device=dev, generator=gen).requires_grad_(True)
    g = torch.randn(B, T, C, device=dev, generator=gen)
    kern = (0.1 * torch.randn(K, device=dev, generator=gen)).requires_grad_(True)
    gain = torch.ones(C, device=dev, requires_grad=True)
    logits = torch.full((1025,), 2.0, device=dev, requires_grad=True)
    gctx = torch.rand(B, C, device=dev, generator=gen)

    def step():
        y = pkg.causal_spectral_conv(x, kern, gain, logits, gctx, None, 32)
        y.backward(g)
        x.grad = kern.grad = gain.grad = logits.grad = None

    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    for _ in range(100):                                  # clock ramp, untimed
        gr.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        gr.replay()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    samples = B * T * C
    out = {"workload": f"fft_lm causal_spectral_conv fwd+bwd (B={B},T={T},C={C},taps={K},n_fft=2048)",
           "steps": steps, "ms_per_step": round(ms, 4), "value": round(samples / ms / 1e6, 2), "unit": "GSamples/s",
           "frac": round(16.0 * samples / (ms * 1e-3) / HBM_PEAK, 4), "launch": "hipGraph, 1 step per replay",
           "algorithmic_bytes_per_sample": 16, "dominant_kernels": "smx::k_conv1<4, 0 / 1, false, 16> (one launch per direction)"}
    try:                                   # HBM bytes of the two launches from the committed PMC passes of this build
        from tensor_cuda_fft_amd import _lib
        tr, src, why = profile_traffic("f2_conv1", lib_sha256(_lib.LIB_PATH), ["smx::k_conv1<4, 0", "smx::k_conv1<4, 1"])
        out["traffic"] = tr
        out["traffic_source"] = src
        out["traffic_note"] = why
    except Exception as e:                                                 # noqa: BLE001
        out["traffic"] = None
        out["traffic_note"] = f"{type(e).__name__}: {e}"
    return out


def measure(rt, args, cfg_name, cfg, steps, custom=False):
    """Warm up, capture, ramp, time EXACTLY `steps` steps of one workload, then time its two transform launches
    with HIP events.  Returns the pieces of the JSON line that belong to this workload."""
    import tensor_cuda_fft_amd as pkg
    from tensor_cuda_fft_amd import _lib, functional

    dev, world, rank = rt.dev, rt.world, rt.rank
    B, N, D, F = cfg["B"], cfg["N"], cfg["D"], cfg["F"]
    module, unit, (w_re, w_im, bias) = make_unit(pkg, cfg, dev, seed=1234)     # replicated weights
    sync = None
    if rt.use_dist:
        pkg.attach_grad_sync(module, mode="overlap" if args.sync_mode == "auto" else args.sync_mode)
        sync = next(m._grad_sync for m in module.modules() if getattr(m, "_grad_sync", None) is not None) \
            if cfg["api"] == "layer" else None
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, N, D, device=dev, generator=gen).requires_grad_(True)
    g = torch.randn(B, N, D, device=dev, generator=gen)
    params = list(module.parameters())

    def step():
        y = unit(x)
        y.backward(g)
        x.grad = None
        for p in params:
            p.grad = None

    sync_all = rt.sync_all

    # warm-up (also builds the twiddle tables and the workspace before any capture)
    n_warm = max(args.warmup, 1)
    for _ in range(n_warm):
        step()
    sync_all()
    steps_before_timing = n_warm

    def capture(n):
        nonlocal steps_before_timing
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        steps_before_timing += 2
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n):
                step()
        return gr

    def time_replays(gr, n):
        """mean seconds per replay over n replays, max over ranks"""
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n):
            gr.replay()
        torch.cuda.synchronize(dev)
        return rt.max_over_ranks((time.perf_counter() - t0) / n)

    # K steps = n_full replays of a graph holding `spg` steps + one graph with the remainder
    plan_runs = []          # (callable, steps it runs)
    launch = args.mode
    sync_report = None
    if args.mode == "graph" and rt.use_dist and rt.backend != "nccl" \
            and os.environ.get("SMX_BENCH_TRY_CAPTURE") != "1":
        launch = f"eager ({rt.backend} collectives cannot be captured)"      # rehearsal backends only
    elif args.mode == "graph":
        try:
            spg = max(1, min(args.steps_per_graph, steps))
            n_full, rem = divmod(steps, spg)
            g_full = capture(spg)
            if sync is not None and args.sync_mode == "auto":
                # both schedules of the collective, 30 replays each after 10 untimed ones; every rank
                # sees the same (max-reduced) numbers, so every rank keeps the same schedule
                trial = {}
                graphs = {"overlap": g_full}
                sync.mode = "fused"
                graphs["fused"] = capture(spg)
                for name, gr in graphs.items():
                    for _ in range(10):
                        gr.replay()
                    trial[name] = time_replays(gr, 30) / spg * 1e3
                    steps_before_timing += 40 * spg
                best = min(trial, key=trial.get)
                sync.mode = best
                g_full = graphs[best]
                sync_report = {"chosen": best, "ms_per_step": {k: round(v, 4) for k, v in trial.items()}}
            plan_runs = [(g_full.replay, spg)] * n_full
            if rem:
                plan_runs.append((capture(rem).replay, rem))
            for _ in range(max(args.warmup // spg, 1)):
                g_full.replay()
                steps_before_timing += spg
            launch = f"hipGraph, {spg} steps per replay"
        except Exception as e:          # e.g. a collective that refuses stream capture
            print(f"[bench] graph capture failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            if os.environ.get("SMX_BENCH_CHILD") == "1":
                os._exit(17)                      # the supervisor repeats the run with eager launches
            # An invalidated capture can leave the streams it touched unusable: continue on fresh ones
            # (a new current stream, a new side stream for the collectives).
            try:
                torch.cuda.synchronize(dev)
            except Exception:
                pass
            if sync is not None:
                sync._side = None
            torch.cuda.set_stream(torch.cuda.Stream(dev))
            launch = "eager (graph capture failed)"
            plan_runs = []
    if not plan_runs:
        plan_runs = [(step, 1)] * steps
        for _ in range(n_warm):
            step()
        steps_before_timing += n_warm

    # ---- clock ramp: same work, untimed (idle -> sustained clocks takes ~0.1 s on MI355X) ------------
    # The number of ramp steps is a pure function of the arguments (NOT of measured time), so every
    # rank issues the same number of collectives.
    est_ms = 16.0 * B * N * D / (0.6 * HBM_PEAK) * 1e3                        # ~step time at 60 %
    pre_steps = int(args.preheat_ms / est_ms)
    done = 0
    i = 0
    while done < pre_steps:
        run, n = plan_runs[i % len(plan_runs)]
        run()
        done += n
        i += 1
        if i % 16 == 0:
            torch.cuda.synchronize(dev)          # keep the launch queue bounded
    torch.cuda.synchronize(dev)
    steps_before_timing += done

    # ---- timed region: EXACTLY K steps between barriers + device syncs ----------------------------
    # (HIP events around every launch call give the per-replay spread; they cost nothing measurable)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan_runs]
    sync_all()
    t0 = time.perf_counter()
    for (run, _), (e0, e1) in zip(plan_runs, evs):
        e0.record()
        run()
        e1.record()
    sync_all()
    dt_local = time.perf_counter() - t0
    dt = rt.max_over_ranks(dt_local)
    per_rank_ms = [round(v / steps * 1e3, 4) for v in rt.gather_over_ranks(dt_local)]
    ms_step = dt / steps * 1e3
    value = world * B * N * D * steps / dt / 1e9
    per_call = [e0.elapsed_time(e1) / n for (e0, e1), (_, n) in zip(evs, plan_runs)]   # ms per step

    # ---- the two transform launches, HIP events on the launch stream ---------------------------------
    # One call of smx_forward / smx_backward (SPECTRUM | INVERSE) with a ready packed filter is exactly
    # the streaming launch(es) of that direction: ONE kernel on the fused plan (c2, c5), the
    # k_split_a / k_split_f / k_split_b group on the residue-split plan (c3).
    plan = _lib.plan(B, N, D, F)
    xd = x.detach()
    reps = max(steps, 20)

    def timed(fn):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(dev)
        ts = [a.elapsed_time(b) for a, b in ev]
        return sum(ts) / len(ts), min(ts)

    with torch.no_grad():
        pack = functional._new_pack(xd, w_re)
        _, xk = functional.forward_raw(xd, w_re, w_im, bias, save_spectrum=True, pack=pack)
        gx = torch.empty_like(g)
        flat = torch.empty(2 * D * F + D, dtype=torch.float32, device=dev)
        f_avg, f_min = timed(lambda: functional.forward_raw(xd, w_re, w_im, bias, save_spectrum=True,
                                                            pack=pack, pack_ready=pack is not None))
        b_avg, b_min = timed(lambda: functional.backward_raw(
            g, xk, w_re, w_im, phases=functional.PHASE_SPECTRUM | functional.PHASE_INVERSE, grad_x=gx,
            flat=flat, pack=pack))
    alg = BYTES_PER_SAMPLE_DIR * B * N * D
    nb = plan.bands
    if plan.nsplit == 1:
        names = {"fwd": f"smx::k_fused<{nb}, 0", "bwd": f"smx::k_fused<{nb}, 1"}
        kind = "one fused launch per direction"
    else:
        names = {"fwd": f"smx::k_split_a<{nb}", "bwd": f"smx::k_split_b<{nb}"}
        kind = (f"launch group per direction (k_split_a + k_split_f + k_split_b, "
                f"{plan.nsplit} residue chunks)")
    launches = {
        "forward": {"kernel": names["fwd"], "avg_ms": round(f_avg, 4), "min_ms": round(f_min, 4),
                    "achieved_GBps": round(alg / (f_avg * 1e-3) / 1e9, 1)},
        "backward": {"kernel": names["bwd"], "avg_ms": round(b_avg, 4), "min_ms": round(b_min, 4),
                     "achieved_GBps": round(alg / (b_avg * 1e-3) / 1e9, 1)},
    }
    dom = "backward" if b_avg >= f_avg else "forward"          # the LONGEST launch of the step
    achieved = launches[dom]["achieved_GBps"]

    sha = lib_sha256(_lib.LIB_PATH)
    traffic_all, traffic_src, traffic_why = (None, None, "custom shape") if custom else \
        profile_traffic(cfg_name, sha, [names["fwd"], names["bwd"]])
    traffic = None
    if traffic_all:
        traffic = traffic_all.get(names["bwd" if dom == "backward" else "fwd"])
    prof_avg, prof_src = (None, "custom shape") if custom else \
        profile_kernel_avg(cfg_name, sha, names["bwd" if dom == "backward" else "fwd"])

    api = 'spectral_mix_with_filter(WirtingerSpectralFilter)' if cfg['api'] == 'wirtinger' else 'SpectralMixingLayer'
    res = {
        "value": round(value, 3), "ms_per_step": round(ms_step, 4),
        "min_ms_per_step": round(min(per_call), 4), "median_ms_per_step": round(statistics.median(per_call), 4),
        "max_ms_per_step": round(max(per_call), 4),
        "per_rank_ms": per_rank_ms,        # every rank's own wall clock over the same K steps (value uses the max)
        "warmup_effective_steps": steps_before_timing,
        "config": {"workload": f"{cfg_name.upper()} {api} fwd+bwd (B={B},N={N},D={D},F={F}) per GPU, fp32, "
                               f"random W/bias/g",
                   "global_batch": B * world,
                   "seq_len": N, "embed_dim": D, "num_filters": F,
                   "parallelism": (f"batch-sharded dp{world}, grad sync {sync.mode if sync else 'n/a'}"
                                   if rt.use_dist else "single GPU"),
                   "grad_sync": (sync.mode if sync else None),
                   "grad_sync_trial": sync_report,
                   "launch": launch, "preheat_ms": args.preheat_ms,
                   "warmup_note": f"--warmup {args.warmup} steps + capture warm-ups + {done} untimed "
                                  f"clock-ramp steps = {steps_before_timing} steps before the timed region",
                   "plan": {"path": plan.path, "L": plan.L, "bands": plan.bands,
                            "nsplit": plan.nsplit, "workgroups": plan.workgroups, "groups": plan.groups}},
        "hbm_roofline_frac_fwd_bwd": round(16.0 * B * N * D / (ms_step * 1e-3) / HBM_PEAK, 4),
        "roofline": {"bound": "hbm", "kernel": f"{launches[dom]['kernel']}> ({dom} launch: the longest of the step; {kind})",
                     "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(achieved * 1e9 / HBM_PEAK, 4), "traffic": traffic,
                     "traffic_source": traffic_src, "traffic_note": traffic_why,
                     "traffic_all_launches": traffic_all,
                     "avg_launch_ms": launches[dom]["avg_ms"], "min_launch_ms": launches[dom]["min_ms"],
                     "avg_launch_source": "HIP events in this process, launches issued one at a time "
                                          "(achieved / frac above)",
                     # the same kernel in the committed rocprofv3 --kernel-trace --stats pass of this build:
                     # average over every launch of the profiled bench command (graph replays, in-step)
                     "profile_avg_launch_ms": None if prof_avg is None else round(prof_avg * 1e-3, 4),
                     "profile_frac": None if prof_avg is None else round(alg / (prof_avg * 1e-6) / HBM_PEAK, 4),
                     "profile_source": prof_src,
                     "algorithmic_bytes_per_launch": alg, "launches": launches,
                     "libsmx_sha256": sha},
    }
    # give the (up to 2 GiB of) tensors of this workload back before the next one is set up
    del x, g, gx, xk, xd, plan_runs, evs, module, unit, params
    functional.release_workspaces()
    torch.cuda.empty_cache()
    return res


def dry_run(rt, args):
    """SMX_BENCH_DRY_RUN=1: the launcher / rendezvous / barrier / max-over-ranks / one-line plumbing with an
    empty step, for machines without a GPU (tests/test_bench_launch.py).  No throughput is reported."""
    if os.environ.get("SMX_BENCH_TEST_FAIL_RANK") == str(rt.rank):          # tests/test_bench_launch.py
        os._exit(3)
    rt.sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    rt.sync_all()
    dt_local = time.perf_counter() - t0
    dt = rt.max_over_ranks(dt_local)
    return {"value": None, "ms_per_step": round(dt / args.steps * 1e3, 6), "dry_run": True,
            "per_rank_ms": [round(v / args.steps * 1e3, 6) for v in rt.gather_over_ranks(dt_local)],
            "config": {"workload": "none (dry run: no GPU work)", "grad_sync": None,
                       "parallelism": f"dp{rt.world}" if rt.use_dist else "single process"}}


def main():
    args = parse_args()
    rt = Runtime(args)
    cfg = dict(CONFIGS[args.config])
    for k, v in (("B", args.batch), ("N", args.seq), ("D", args.dim), ("F", args.filters)):
        if v:
            cfg[k] = v
    if args.dim and not args.filters:
        cfg["F"] = args.dim // 2
    custom = any((args.batch, args.seq, args.dim, args.filters))

    if args.opts and not rt.dry:
        from tensor_cuda_fft_amd import _lib
        for kv in filter(None, args.opts.split(";")):
            k, v = kv.split("=")
            _lib.set_option(k.strip(), int(v))
    res = dry_run(rt, args) if rt.dry else measure(rt, args, args.config, cfg, args.steps, custom)
    if args.opts:
        res.setdefault("config", {})["opts"] = args.opts
    out = {
        "metric": f"spectral-mix fwd+bwd GSamples/s (B*N*D/s) at N={cfg['N']},D={cfg['D']}; %HBM roofline",
        "value": res.pop("value"), "unit": "GSamples/s", "n_gpus": rt.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # what the collective library saw (SUM all-reduce of 1.0 per rank) and over which backend
        "ranks_seen": rt.ranks_seen,
        "collective_backend": (("rccl (torch backend nccl)" if rt.backend == "nccl" else rt.backend)
                               if rt.use_dist else None),
        "launched_by": os.environ.get("SMX_BENCH_LAUNCHED_BY", "direct" if rt.world == 1 else "external launcher"),
    }
    out.update(res)

    # ---- the other single-GPU BASELINE configs on the same line (same binary, same box, same protocol) ------
    if (rt.world == 1 and not rt.dry and not custom and args.config == "c2" and not args.no_other_configs
            and not rt.use_dist):
        others = {}
        for name in ("c3", "c5"):
            try:
                r = measure(rt, args, name, dict(CONFIGS[name]), args.steps)
                rf = r["roofline"]
                others[name] = {
                    "workload": r["config"]["workload"], "steps": args.steps,
                    "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "GSamples/s",
                    "frac": r["hbm_roofline_frac_fwd_bwd"],
                    "min_ms_per_step": r["min_ms_per_step"], "max_ms_per_step": r["max_ms_per_step"],
                    "dominant_kernel": rf["kernel"], "dominant_kernel_frac": rf["frac"],
                    "dominant_kernel_avg_ms": rf["avg_launch_ms"], "launches": rf["launches"],
                    "traffic": rf["traffic"], "traffic_source": rf["traffic_source"],
                    "plan": r["config"]["plan"], "launch": r["config"]["launch"]}
            except Exception as e:                                         # noqa: BLE001
                others[name] = {"error": f"{type(e).__name__}: {e}"}
        try:                                   # ... and the 8(f) row f2 (fft_lm's causal convolution), same box
            others["f2"] = measure_f2(rt, args.steps)
        except Exception as e:                                             # noqa: BLE001
            others["f2"] = {"error": f"{type(e).__name__}: {e}"}
        out["other_configs"] = others

    if rt.rank == 0 and not rt.dry and os.environ.get("SMX_BENCH_NO_BOX") != "1":
        try:
            out["box"] = measure_box(rt.dev)
            out["box"]["step_over_copy"] = round(out["ms_per_step"] * 1e3 / out["box"]["copy_256MiB_us"], 3)
        except Exception as e:                                             # noqa: BLE001
            out["box"] = {"error": f"{type(e).__name__}: {e}"}
    if rt.rank == 0 and rt.world == 1 and not rt.dry and not args.no_cpu_baseline:
        iters = 8 if args.config == "c2" else 3
        out["cpu_baseline"] = cpu_baseline(cfg["B"], cfg["N"], cfg["D"], cfg["F"], iters)
    if rt.rank == 0:
        print(json.dumps(out), flush=True)
    rt.close()


def supervise():
    """N > 1 only.  A failed hipGraph capture (a collective that refuses stream capture) leaves this HIP
    stack unusable for the rest of the process -- later launches fail or crash -- so the graph-mode attempt
    runs in a child process and, if that child fails, a second child repeats the run with eager launches on
    a fresh rendezvous port.  This parent never touches the GPU; stdout/stderr are inherited."""
    import subprocess

    def run(extra, env):
        try:
            return subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra, env=env,
                                   timeout=ATTEMPT_S)
        except subprocess.TimeoutExpired:
            return 124

    env = dict(os.environ, SMX_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: what RCCL needs on this driver
    rc = run([], env)
    if rc not in (0, EXIT_NO_DEVICE) and "eager" not in sys.argv:
        print(f"[bench] graph-mode run exited with {rc}; repeating with eager launches", file=sys.stderr,
              flush=True)
        # a rendezvous of its own: rank 0's child hosts a new store on another port (the launcher's agent
        # store still holds the keys of the first attempt)
        env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29531")) + 7)
        env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
        rc = run(["--mode", "eager"], env)
    sys.exit(rc)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n):
    """`python3 bench.py --gpus N` without a launcher: this parent -- which never touches the GPU -- starts N
    fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, each its own session),
    waits for all of them and passes rank 0's single JSON line on.  If the graph-mode attempt fails in any rank
    (exit 17 = a collective refused stream capture; see supervise()) every rank is started again with eager
    launches on a fresh rendezvous port.  Exit status: 0 only if every rank of the last attempt exited 0."""
    import signal
    import subprocess
    import tempfile

    def attempt(extra):
        port = _free_port()
        procs = []
        outs = []
        errs = []
        timed_out = False
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMX_BENCH_CHILD="1",
                       SMX_BENCH_LAUNCHED_BY="bench.py launch_ranks")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
            out = tempfile.TemporaryFile(mode="w+") if r == 0 else None
            outs.append(out)
            err = tempfile.TemporaryFile(mode="w+")               # every rank's stderr: its tail is shown if it fails
            errs.append(err)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra,
                                          env=env, stdout=out if out is not None else err, stderr=err,
                                          start_new_session=True))
        deadline = time.monotonic() + ATTEMPT_S
        grace = None                     # once one rank has failed the others get 20 s to follow
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            now = time.monotonic()
            if grace is None and any(p.poll() not in (None, 0) for p in procs):
                grace = now + 20
            if now > deadline or (grace is not None and now > grace):
                timed_out = now > deadline
                for p in procs:
                    if p.poll() is None:
                        try:
                            os.killpg(p.pid, signal.SIGKILL)      # the session we started, nothing else
                        except ProcessLookupError:
                            pass
                break
        rcs = [p.wait() for p in procs]
        outs[0].seek(0)
        text = outs[0].read()
        outs[0].close()
        if timed_out:
            print(f"[bench] attempt {extra or ['graph']} hit its {ATTEMPT_S:.0f} s limit: ranks still running were "
                  f"killed (exit codes {rcs})", file=sys.stderr, flush=True)
        for r, e in enumerate(errs):                              # a failed attempt is legible: who said what last
            e.seek(0)
            lines = e.read().splitlines()
            e.close()
            if any(rcs):
                print(f"[bench] rank {r} exit {rcs[r]}; last stderr lines:", file=sys.stderr, flush=True)
                for ln in lines[-12:]:
                    print(f"[bench]   r{r}| {ln}", file=sys.stderr, flush=True)
            elif r == 0:
                for ln in lines:
                    print(ln, file=sys.stderr, flush=True)
        return rcs, text

    rcs, text = attempt([])
    if any(rcs) and EXIT_NO_DEVICE not in rcs and "eager" not in sys.argv:
        print(f"[bench] graph-mode attempt: rank exit codes {rcs}; repeating with eager launches",
              file=sys.stderr, flush=True)
        rcs, text = attempt(["--mode", "eager"])
    # stdout carries the JSON line and nothing else (gloo's C++ side prints connection notes to stdout)
    for line in text.splitlines():
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr, flush=True)
        sys.exit(next(rc for rc in rcs if rc) & 0xFF or 1)
    sys.exit(0)


if __name__ == "__main__":
    n_arg = parse_args().gpus
    if "WORLD_SIZE" not in os.environ and n_arg > 1:
        launch_ranks(n_arg)                       # no launcher: be the launcher
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if multi and os.environ.get("SMX_BENCH_CHILD") != "1" and "--no-supervise" not in sys.argv:
        supervise()                               # under torch.distributed.run: one supervised child per rank
    main()
