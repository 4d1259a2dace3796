#!/usr/bin/env python3
"""Random-shape fuzz of the general entry points against fp64 references (run on the GPU box):
spectral_filter (zero-padded rows, explicit k incl. Nyquist, optional row_scale) on every plan -- fused 1/2/4
bands, residue split, four-step (L = 5..16, 32), band groups, direct literal / matrix-core tiles --,
rank_one_conv, seq_fft, the transform pair rfft / irfft, and the twin blocks' gate chain and fusion line.  Prints failing cases; exit code = number of failures."""
import argparse, os, random, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensor_cuda_fft_amd import _lib, functional as fn
from oracle import spectral_oracle as so

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rnd = random.Random(args.seed)
dev = torch.device("cuda:0")
T = torch.from_numpy


def rel(a, r, floor=0.0):
    a = np.asarray(a, np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(r) else np.float64)
    r = np.asarray(r)
    return float(np.abs(a - r).max() / max(np.abs(r).max(), floor, 1e-30))


bad = 0
for case in range(args.cases):
    kind = rnd.choice(["filter", "filter", "filter", "conv", "cfft", "pair", "pair", "gate", "mix"])
    if kind == "filter":
        L = rnd.choice([1, 1, 2, 3, 4, 5, 6, 8, 8, 12, 16, 17, 20, 22, 30, 32, 48, 64, 128,
                        19, 23, 27, 31, 36, 40, 44, 52, 56, 60, 72, 80, 88, 104, 120, 144, 176, 208, 240, 34])
        n_fft = 256 * L if rnd.random() < 0.85 else rnd.choice([96, 200, 333, 1000, 1500])
        R = n_fft if rnd.random() < 0.5 else rnd.randint(max(1, n_fft // 3), n_fft)
        D = rnd.choice([2, 4, 6, 10, 34, 64, 90, 7, 33])
        B = rnd.choice([1, 2, 3, 5, 17])
        kmax = n_fft // 2 + 1
        k = rnd.choice([kmax, kmax, kmax - 1, rnd.randint(1, kmax), min(kmax, 128), min(kmax, 600)])
        F = k + rnd.choice([0, 0, 3])
        if B * R * D * min(k, 2048) > 3e8 or B * n_fft * D > 6e6:
            continue
        _lib.set_option("nsplit", rnd.choice([0, 0, 0, 2, 1 << 20]))
        _lib.set_option("fourstep", rnd.choice([1, 1, 1, 0]))
        use_sc = rnd.random() < 0.4
        use_b = (not use_sc) and rnd.random() < 0.5
        rng = np.random.default_rng(case)
        x = rng.standard_normal((B, R, D)).astype(np.float32); g = rng.standard_normal((B, R, D)).astype(np.float32)
        wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
        wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
        bias = (0.1 * rng.standard_normal(D)).astype(np.float32) if use_b else None
        sc = (0.5 + rng.random((B, D))).astype(np.float32) if use_sc else None
        leaves = [T(a).to(dev).requires_grad_(True) for a in (x, wr, wi)]
        bd = T(bias).to(dev).requires_grad_(True) if use_b else None
        sd = T(sc).to(dev).requires_grad_(True) if use_sc else None
        try:
            y = fn.spectral_filter(leaves[0], leaves[1], leaves[2], bd, n_fft=n_fft, k=k, row_scale=sd)
            y.backward(T(g).to(dev))
            torch.cuda.synchronize()
        except Exception as e:                                     # noqa: BLE001
            print("EXC", case, (B, R, D, F, n_fft, k), type(e).__name__, e, flush=True); bad += 1; continue
        y0, _ = so.forward_closed_ex(x, wr, wi, bias, n_fft, k)
        s3 = 1.0 if sc is None else sc[:, None, :]
        gx, gwr, gwi, gb = so.backward_closed_ex(x, wr, wi, g * s3, n_fft, k)
        e = {"y": rel(y.detach().cpu().numpy(), y0 * s3), "gx": rel(leaves[0].grad.cpu().numpy(), gx)}
        floor = 1e-3 * float(np.abs(gwr).max())
        e["gwr"] = rel(leaves[1].grad.cpu().numpy(), gwr, floor)
        e["gwi"] = rel(leaves[2].grad.cpu().numpy(), gwi, floor)
        if use_b:
            e["gb"] = rel(bd.grad.cpu().numpy(), gb)
        if use_sc:
            e["gs"] = rel(sd.grad.cpu().numpy(), (g.astype(np.float64) * y0).sum(axis=1))
        ok = e["y"] <= 1e-5 and e["gx"] <= 1e-5 and max(v for kk, v in e.items() if kk not in ("y", "gx")) <= 1e-4
        p = _lib.plan_ex(_lib.smx_shape(B, R, D, F, n_fft, k))
        tag = f"filter B={B} R={R} D={D} F={F} n={n_fft} k={k} sc={use_sc} path={p.path} bands={p.bands} groups={p.groups} nsplit={p.nsplit}"
    elif kind == "conv":
        _lib.set_option("fourstep", 1); _lib.set_option("nsplit", 0)
        n_fft = rnd.choice([512, 1024, 2048, 2048, 4096, 8192, 16384, 32768])
        R = rnd.randint(n_fft // 4, n_fft)
        D = rnd.choice([2, 6, 34, 64, 90])
        B = rnd.choice([1, 2, 5])
        if B * n_fft * D > 4e6:
            continue
        rng = np.random.default_rng(case)
        fb = n_fft // 2 + 1
        x = rng.standard_normal((B, R, D)).astype(np.float32); g = rng.standard_normal((B, R, D)).astype(np.float32)
        hr = rng.standard_normal(fb).astype(np.float32); hi = rng.standard_normal(fb).astype(np.float32)
        sc = (0.5 + rng.random((B, D))).astype(np.float32)
        xd, hrd, hid, scd = (T(a).to(dev).requires_grad_(True) for a in (x, hr, hi, sc))
        y = fn.rank_one_conv(xd, hrd, hid, scd, n_fft)
        y.backward(T(g).to(dev)); torch.cuda.synchronize()
        xt, hrt, hit, sct = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, hr, hi, sc))
        X = torch.fft.rfft(torch.nn.functional.pad(xt, (0, 0, 0, n_fft - R)), dim=1)
        yr = torch.fft.irfft(X * torch.complex(hrt, hit)[None, :, None], n=n_fft, dim=1)[:, :R] * sct[:, None, :]
        yr.backward(torch.tensor(g, dtype=torch.float64))
        e = {"y": rel(y.detach().cpu().numpy(), yr.detach().numpy()), "gx": rel(xd.grad.cpu().numpy(), xt.grad.numpy()),
             "gs": rel(scd.grad.cpu().numpy(), sct.grad.numpy()), "ghr": rel(hrd.grad.cpu().numpy(), hrt.grad.numpy()),
             "ghi": rel(hid.grad.cpu().numpy(), hit.grad.numpy())}
        ok = e["y"] <= 1e-5 and e["gx"] <= 1e-5 and max(e["gs"], e["ghr"], e["ghi"]) <= 1e-4
        tag = f"conv B={B} R={R} D={D} n={n_fft}"
    elif kind == "pair":
        # functional.rfft / irfft: values and gradients against torch.fft in float64, every plan
        L = rnd.choice([1, 1, 2, 3, 4, 5, 6, 8, 8, 12, 16, 17, 20, 22, 30, 32, 48, 64, 128,
                        19, 23, 27, 31, 36, 40, 44, 52, 56, 60, 72, 80, 88, 104, 120, 144, 176, 208, 240, 34])
        n_fft = 256 * L if rnd.random() < 0.85 else rnd.choice([96, 200, 333, 1000, 1500])
        R = n_fft if rnd.random() < 0.5 else rnd.randint(max(1, n_fft // 3), n_fft)
        D = rnd.choice([2, 4, 6, 10, 34, 64, 90, 7, 33])
        B = rnd.choice([1, 2, 3, 5, 17])
        kmax = n_fft // 2 + 1
        k = rnd.choice([kmax, kmax, kmax, kmax - 1, rnd.randint(1, kmax), min(kmax, 128), min(kmax, 600)])
        if B * R * D * min(k, 2048) > 3e8 or B * n_fft * D > 6e6:
            continue
        _lib.set_option("nsplit", rnd.choice([0, 0, 0, 2, 1 << 20]))
        _lib.set_option("fourstep", rnd.choice([1, 1, 1, 0]))
        rng = np.random.default_rng(case)
        x = rng.standard_normal((B, R, D)).astype(np.float32)
        sp = (rng.standard_normal((B, k, D)) + 1j * rng.standard_normal((B, k, D))).astype(np.complex64)
        gs = (rng.standard_normal((B, k, D)) + 1j * rng.standard_normal((B, k, D))).astype(np.complex64)
        gy = rng.standard_normal((B, R, D)).astype(np.float32)
        xd = T(x).to(dev).requires_grad_(True); sd_ = T(sp).to(dev).requires_grad_(True)
        try:
            X = fn.rfft(xd, n_fft, k); X.backward(T(gs).to(dev))
            y = fn.irfft(sd_, n_fft, R); y.backward(T(gy).to(dev))
            torch.cuda.synchronize()
        except Exception as ex:                                    # noqa: BLE001
            print("EXC", case, (B, R, D, n_fft, k), type(ex).__name__, ex, flush=True); bad += 1; continue
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        st = torch.tensor(sp, dtype=torch.complex128, requires_grad=True)
        Xr = torch.fft.rfft(xt, n=n_fft, dim=1)[:, :k]; Xr.backward(torch.tensor(gs, dtype=torch.complex128))
        yr = torch.fft.irfft(st, n=n_fft, dim=1)[:, :R]; yr.backward(torch.tensor(gy, dtype=torch.float64))
        e = {"X": rel(X.detach().cpu().numpy(), Xr.detach().numpy()), "gx": rel(xd.grad.cpu().numpy(), xt.grad.numpy()),
             "y": rel(y.detach().cpu().numpy(), yr.detach().numpy()), "gS": rel(sd_.grad.cpu().numpy(), st.grad.numpy())}
        ok = max(e.values()) <= 1e-5
        p = _lib.plan_ex(_lib.smx_shape(B, R, D, max(k, 1), n_fft, k))
        tag = f"pair B={B} R={R} D={D} n={n_fft} k={k} path={p.path} bands={p.bands} groups={p.groups} nsplit={p.nsplit}"
    elif kind == "gate":
        # the gate chain of the twin blocks (smx_spectral_gate_*) against fp64 autograd of the reference lines
        # (fft_lm/frequency_native.py:95, :338, :351) with the oracle's port of the hand-written backward
        B = rnd.choice([1, 2, 3, 7]); Fq = rnd.choice([1, 2, 9, 33, 65, 130, 257, 1025]); C = 2 * rnd.choice([1, 3, 8, 33, 64, 65, 200])
        if B * Fq * C > 2e6:
            continue
        torch.manual_seed(case)
        quirk = rnd.random() < 0.5
        x = torch.randn(B, Fq, C, dtype=torch.complex64, device=dev); g = torch.randn_like(x)
        a = torch.randn(Fq, dtype=torch.complex64, device=dev)
        u = 1 + 0.3 * torch.randn(C, device=dev) if quirk or rnd.random() < 0.5 else None
        p = torch.rand(Fq, device=dev) if rnd.random() < 0.6 else None
        q = torch.rand(B, C, device=dev) if rnd.random() < 0.6 else None
        m = (torch.rand(Fq, device=dev) > 0.3).float() if rnd.random() < 0.6 else None
        lv = [t.clone().requires_grad_(True) if t is not None else None for t in (x, a, u, p, q)]
        yg = fn.spectral_gate(lv[0], lv[1], lv[2], lv[3], lv[4], m, reference_gain_grad=quirk)
        yg.backward(g)
        d64 = lambda t: None if t is None else t.detach().to(torch.complex128 if t.is_complex() else torch.float64).cpu().requires_grad_(True)
        rl = [d64(t) for t in (x, a, u, p, q)]
        if quirk:
            class Conv(torch.autograd.Function):
                @staticmethod
                def forward(ctx, xf, kf, gain):
                    ctx.save_for_backward(xf, kf, gain)
                    return so.freqconv_port(xf, kf, gain, xf)[0]
                @staticmethod
                def backward(ctx, go):
                    return so.freqconv_port(*ctx.saved_tensors, go)[1:]
            r = Conv.apply(rl[0], rl[1], rl[2])
        else:
            r = rl[0] * rl[1].view(1, -1, 1)
            if rl[2] is not None:
                r = r * rl[2].view(1, 1, -1)
        for f_ in (rl[3].view(1, -1, 1) if rl[3] is not None else None, rl[4].unsqueeze(1) if rl[4] is not None else None,
                   m.double().cpu().view(1, -1, 1) if m is not None else None):
            if f_ is not None:
                r = r * f_
        r.backward(g.to(torch.complex128).cpu())
        e = {n: rel(t.grad.cpu().numpy(), rt.grad.numpy()) for n, t, rt in zip(("gx", "ga", "gu", "gp", "gq"), lv, rl) if t is not None}
        e["y"] = rel(yg.detach().cpu().numpy(), r.detach().numpy())
        ok = all(v <= (2e-6 if n in ("gx", "y") else 2e-5) for n, v in e.items())
        tag = f"gate B={B} F={Fq} C={C} quirk={quirk} u={u is not None} p={p is not None} q={q is not None} m={m is not None}"
    elif kind == "mix":
        n_el = 4 * rnd.choice([1, 3, 64, 1000, 4097, 250000])
        torch.manual_seed(case)
        ts = [torch.randn(n_el, device=dev) for _ in range(4)]
        if rnd.random() < 0.3:
            ts[3] = None
        w = torch.rand(2, device=dev); g = torch.randn(n_el, device=dev)
        lv = [t.clone().requires_grad_(True) if t is not None else None for t in ts + [w]]
        fn.mix_paths(lv[0], lv[1], lv[2], lv[3], lv[4], 0.1).backward(g)
        rl = [t.detach().double().cpu().requires_grad_(True) if t is not None else None for t in ts + [w]]
        r = rl[0] + rl[4][0] * rl[1] + rl[4][1] * rl[2] + (0.1 * rl[3] if rl[3] is not None else 0)
        r.backward(g.double().cpu())
        e = {n: rel(t.grad.cpu().numpy(), rt.grad.numpy()) for n, t, rt in zip(("gr", "ga", "gb", "gc", "gw"), lv, rl) if t is not None}
        ok = all(v <= (2e-6 if n != "gw" else 2e-5) for n, v in e.items())
        tag = f"mix n={n_el} c={ts[3] is not None}"
    else:
        _lib.set_option("fourstep", 1); _lib.set_option("nsplit", 0)
        N = rnd.choice([256 * rnd.choice([1, 2, 3, 4, 5, 7, 8, 16, 17, 20, 26, 32, 64, 128, 21, 29, 36, 44, 48, 60, 80, 112, 144, 240]),
                        rnd.choice([30, 100, 333])])
        D = rnd.choice([1, 3, 8, 40]); B = rnd.choice([1, 2, 4])
        if B * N * D > 2e6:
            continue
        rng = np.random.default_rng(case)
        z = (rng.standard_normal((B, N, D)) + 1j * rng.standard_normal((B, N, D))).astype(np.complex64)
        out = fn.seq_fft_raw(T(z).to(dev))
        e = {"Z": rel(out.cpu().numpy(), np.fft.fft(z.astype(np.complex128), axis=1))}
        ok = e["Z"] <= 1e-5
        tag = f"cfft B={B} N={N} D={D}"
    if not ok:
        bad += 1
        print("FAIL", case, tag, {kk: f"{v:.2e}" for kk, v in e.items()}, flush=True)
    elif case % 25 == 0:
        print("ok", case, tag, {kk: f"{v:.1e}" for kk, v in e.items()}, flush=True)
_lib.set_option("nsplit", 0); _lib.set_option("fourstep", 1)
print(f"done: {bad} failures")
sys.exit(min(bad, 100))
