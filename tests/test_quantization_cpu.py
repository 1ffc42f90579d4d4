"""CPU tests of the native PTQ calibrators."""
import math

import torch

from bevformer_tensorrt_amd import quantization as Q


def test_minmax_tracks_running_max():
    c = Q.get_calibrator("minmax")()
    c.collect("a", torch.tensor([0.5, -2.0]))
    c.collect("a", torch.tensor([1.0, 3.0]))
    assert abs(c.scale("a") - 3.0 / 127) < 1e-9
    q = c.quantize(torch.tensor([3.0, -3.0, 100.0, 0.011]), c.scale("a"))
    assert q.tolist() == [127, -127, 127, 0]


def test_histogram_range_growth_keeps_counts():
    c = Q.PercentileCalibrator(100.0)
    g = torch.Generator().manual_seed(0)
    c.collect("x", torch.rand(10000, generator=g))            # range ~1
    c.collect("x", torch.rand(10000, generator=g) * 7.5)      # forces 3 doublings
    st = c._stats["x"]
    assert st["range"] >= 7.5 and abs(float(st["hist"].sum()) - 20000) < 1e-6
    assert abs(c.scale("x") * 127 - 7.5) < 0.05


def test_entropy_clips_gaussian_tail_but_keeps_uniform():
    g = torch.Generator().manual_seed(0)
    ce = Q.get_calibrator("entropy")()
    cm = Q.get_calibrator("minmax")()
    x = torch.randn(400000, generator=g)
    x[0] = 40.0                                                  # one far outlier
    for t in x.split(100000):
        ce.collect("g", t)
        cm.collect("g", t)
    thr_e, thr_m = ce.scale("g") * 127, cm.scale("g") * 127
    assert thr_m == 40.0
    assert 2.5 < thr_e < 8.0, thr_e                              # outlier ignored, bulk kept
    # a uniform distribution has no tail to clip: threshold ~ the max
    u = torch.rand(200000, generator=g) * 2 - 1
    ce.collect("u", u)
    assert 0.9 < ce.scale("u") * 127 <= 1.01


def test_entropy_threshold_minimises_kl_against_brute_force():
    g = torch.Generator().manual_seed(1)
    hist = torch.histc(torch.randn(50000, generator=g).abs(), bins=2048, min=0, max=6.0).double()
    i = Q.entropy_threshold_bin(hist)

    def kl_at(i):   # independent, loop-based statement of the same objective
        p = hist[:i].clone()
        p[i - 1] += hist[i:].sum()
        q = torch.zeros(i, dtype=torch.float64)
        for lv in range(128):
            lo, hi = int(math.ceil(lv * i / 128 - 0.5 + 1e-9)), int(math.ceil((lv + 1) * i / 128 - 0.5 + 1e-9))
            lo, hi = max(lo, 0), min(hi, i)
            seg = hist[lo:hi]
            nz = (seg > 0).sum()
            if nz > 0:
                q[lo:hi] = torch.where(seg > 0, seg.sum() / nz, torch.zeros_like(seg))
        pn, qn = p / p.sum(), q / q.sum()
        m = pn > 0
        return float((pn[m] * torch.log(pn[m] / qn[m].clamp(min=1e-12))).sum())

    best = min(range(128, 2049, 16), key=kl_at)
    assert abs(kl_at(i + 1) - kl_at(best)) <= 0.02 * max(kl_at(best), 1e-6) + 1e-4
