#!/usr/bin/env python3
"""Random shapes through the one-launch rank-one convolution (k_conv1, option conv1 = 2) against float64 autograd of
the reference's op sequence (fft_lm/train_fixed_full.py:515-555) and against the three-launch form of the same
library (conv1 = 0); smx_conv_response against float64 autograd as well.  Prints failing cases; exit code = number
of failures.  Test infrastructure (run on the GPU box): python tests/tools/fuzz_conv1_gpu.py --cases 300"""
import argparse, os, random, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tensor_cuda_fft_amd as pkg                      # noqa: E402,F401
from tensor_cuda_fft_amd import _lib, functional as fn   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rnd = random.Random(args.seed)
dev = torch.device("cuda:0")
T = torch.from_numpy


def rel(a, r, floor=0.0):
    a = np.asarray(a, np.float64); r = np.asarray(r, np.float64)
    return float(np.abs(a - r).max() / max(np.abs(r).max(), floor, 1e-30))


bad = 0
for case in range(args.cases):
    n_fft = rnd.choice([512, 1024, 2048, 2048])
    R = rnd.choice([1, 2, 15, 16, 17, n_fft // 2, n_fft // 2 - 1, rnd.randint(1, n_fft // 2), rnd.randint(1, n_fft // 2),
                    n_fft // 2 + 1, n_fft, n_fft - 1, rnd.randint(n_fft // 2 + 1, n_fft), rnd.randint(n_fft // 2 + 1, n_fft)])
    D = rnd.choice([2, 4, 6, 30, 32, 34, 62, 64, 66, 90, 128])
    B = rnd.choice([1, 2, 3, 5, 8, 9])
    use_scale = rnd.random() < 0.8
    rng = np.random.default_rng(1000 + case)
    fb = n_fft // 2 + 1
    x = rng.standard_normal((B, R, D)).astype(np.float32); g = rng.standard_normal((B, R, D)).astype(np.float32)
    hr = rng.standard_normal(fb).astype(np.float32); hi = rng.standard_normal(fb).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32) if use_scale else None

    def run(conv1):
        with _lib.options(conv1=conv1):
            xd, hrd, hid = (T(a).to(dev).requires_grad_(True) for a in (x, hr, hi))
            scd = None if sc is None else T(sc).to(dev).requires_grad_(True)
            y = fn.rank_one_conv(xd, hrd, hid, scd, n_fft)
            y.backward(T(g).to(dev)); torch.cuda.synchronize()
            out = [y.detach().cpu().numpy(), xd.grad.cpu().numpy(), hrd.grad.cpu().numpy(), hid.grad.cpu().numpy()]
            if scd is not None:
                out.append(scd.grad.cpu().numpy())
            return out

    new, old = run(rnd.choice([2, 3])), run(0)          # 2: 512-thread workgroups, 3: 256-thread workgroups on 16 channels
    xt, hrt, hit = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, hr, hi))
    sct = None if sc is None else torch.tensor(sc, dtype=torch.float64, requires_grad=True)
    X = torch.fft.rfft(torch.nn.functional.pad(xt, (0, 0, 0, n_fft - R)), dim=1)
    yr = torch.fft.irfft(X * torch.complex(hrt, hit)[None, :, None], n=n_fft, dim=1)[:, :R]
    if sct is not None:
        yr = yr * sct[:, None, :]
    yr.backward(torch.tensor(g, dtype=torch.float64))
    ref = [yr.detach().numpy(), xt.grad.numpy(), hrt.grad.numpy(), hit.grad.numpy()] + \
          ([] if sct is None else [sct.grad.numpy()])
    tols = [1e-5, 1e-5, 1e-4, 1e-4, 1e-4]
    # natural scale of an output row: |x| |H| sqrt(bins) / n_fft -- a one-row input makes y = x[0] * mean(H), a sum of a
    # thousand random bins that cancels to a tenth of that, and the error is relative to what was summed
    nat = float(np.abs(x).max() * np.hypot(hr, hi).max() * np.sqrt(fb) / n_fft) * (1.0 if sc is None else float(sc.max()))
    floors = [nat, nat * float(np.abs(g).max()) / max(float(np.abs(x).max()), 1e-30), 0.0, 0.0, 0.0]
    errs = [max(rel(a, r, fl), rel(a, b, fl)) for a, b, r, fl in zip(new, old, ref, floors)]
    ok = all(e <= t for e, t in zip(errs, tols))
    # the response kernel on a random (n_fft, taps, logits length, mask)
    K = rnd.choice([1, 7, 64, 128, min(200, n_fft)])
    nl = fb + rnd.choice([0, 0, 3, 100])
    k = (0.3 * rng.standard_normal(K)).astype(np.float32); lg = rng.standard_normal(nl).astype(np.float32)
    m = rng.random(fb).astype(np.float32) if rnd.random() < 0.5 else None
    kd, ld = T(k).to(dev).requires_grad_(True), T(lg).to(dev).requires_grad_(True)
    h0, h1 = fn.conv_response(kd, ld, None if m is None else T(m).to(dev), n_fft)
    (h0 * T(hr).to(dev) + h1 * T(hi).to(dev)).sum().backward()
    kt = torch.tensor(k, dtype=torch.float64, requires_grad=True); lt = torch.tensor(lg, dtype=torch.float64, requires_grad=True)
    H = torch.fft.rfft(torch.nn.functional.pad(kt, (0, n_fft - K))) * torch.sigmoid(lt[:fb])
    if m is not None:
        H = H * torch.tensor(m, dtype=torch.float64)
    (H.real * torch.tensor(hr, dtype=torch.float64) + H.imag * torch.tensor(hi, dtype=torch.float64)).sum().backward()
    er = [rel(h0.detach().cpu().numpy(), H.real.detach().numpy()), rel(h1.detach().cpu().numpy(), H.imag.detach().numpy()),
          rel(kd.grad.cpu().numpy(), kt.grad.numpy()), rel(ld.grad.cpu().numpy(), lt.grad.numpy())]
    ok = ok and er[0] <= 1e-5 and er[1] <= 1e-5 and er[2] <= 1e-4 and er[3] <= 1e-4
    if not ok:
        bad += 1
        print(f"FAIL case {case}: conv B={B} R={R} D={D} n={n_fft} scale={use_scale} errs={errs}; response K={K} nl={nl} errs={er}",
              flush=True)
    if case % 50 == 49:
        print(f"[fuzz_conv1] {case + 1} cases, {bad} failures", flush=True)
print(f"fuzz_conv1_gpu: {args.cases} cases, {bad} failures")
sys.exit(min(bad, 100))
