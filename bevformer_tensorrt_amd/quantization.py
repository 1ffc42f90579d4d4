"""Per-tensor PTQ calibrators producing the scales the INT8 operators take.

In the reference the scales come from TensorRT: `get_calibrator("minmax" | "entropy" | "legacy")`
wraps `trt.IInt8MinMaxCalibrator` / `IInt8EntropyCalibrator2` / `IInt8LegacyCalibrator`
(det2trt/quantization/calibrator_trt.py:6-92) and the engine hands each plugin its tensors'
`PluginTensorDesc::scale` (multiScaleDeformableAttnPlugin.cpp:75-77); the QDQ flow uses
pytorch_quantization's max / 99.99-percentile histogram calibrators
(det2trt/quantization/calibrator_qdq.py:29-80).  Neither library exists on ROCm, so the
published algorithms are restated here as host logic (torch tensors, histograms on the
device the data lives on):

  * MinMaxCalibrator      scale = max|x| / 127 over all calibration batches
  * PercentileCalibrator  scale = percentile(|x|, p) / 127 from a running histogram (p = 99.99)
  * EntropyCalibrator     TensorRT "entropy calibration 2" (8-bit Inference with TensorRT,
                          S. Migacz, GTC 2017): 2048-bin histogram of |x|; for every candidate
                          clip bin i in [128, 2048): reference P = hist[:i] with the outliers
                          folded into the last bin, candidate Q = P merged into 128 levels and
                          expanded back over P's non-empty bins; threshold = argmin KL(P || Q).

`get_calibrator(name)` mirrors the reference's factory; `Calibrator.collect(name, tensor)` is
called per boundary tensor per calibration frame; `scales()` returns {name: scale}.
"""
import math

import torch

_NUM_BINS = 2048
_NUM_LEVELS = 128


class _Base:
    def __init__(self):
        self._stats = {}

    def collect(self, name, tensor):
        raise NotImplementedError

    def scale(self, name):
        raise NotImplementedError

    def scales(self):
        return {k: self.scale(k) for k in self._stats}

    @staticmethod
    def quantize(tensor, scale):
        """real -> int8 with round-to-nearest-even and saturation to [-127, 127]."""
        return torch.clamp(torch.round(tensor.float() / scale), -127, 127).to(torch.int8)


class MinMaxCalibrator(_Base):
    def collect(self, name, tensor):
        m = float(tensor.detach().abs().max())
        self._stats[name] = max(self._stats.get(name, 0.0), m)

    def scale(self, name):
        return max(self._stats[name], 1e-12) / 127.0


class _Histogram(_Base):
    """Running histogram of |x| with a range that doubles when a batch exceeds it (bins merge
    pairwise, so earlier batches stay exactly accounted for)."""

    def collect(self, name, tensor):
        x = tensor.detach().abs().float().flatten()
        amax = float(x.max())
        st = self._stats.get(name)
        if st is None:
            st = {"hist": torch.zeros(_NUM_BINS, dtype=torch.float64), "range": max(amax, 1e-12)}
            self._stats[name] = st
        while amax > st["range"]:
            st["hist"] = torch.cat([st["hist"].view(-1, 2).sum(1), torch.zeros(_NUM_BINS // 2, dtype=torch.float64)])
            st["range"] *= 2
        h = torch.histc(x, bins=_NUM_BINS, min=0.0, max=st["range"])
        st["hist"] += h.double().cpu()


class PercentileCalibrator(_Histogram):
    def __init__(self, percentile=99.99):
        super().__init__()
        self.percentile = percentile

    def scale(self, name):
        st = self._stats[name]
        cdf = torch.cumsum(st["hist"], 0)
        idx = int(torch.searchsorted(cdf, cdf[-1] * self.percentile / 100.0))
        return (min(idx, _NUM_BINS - 1) + 0.5) * st["range"] / _NUM_BINS / 127.0


class EntropyCalibrator(_Histogram):
    def scale(self, name):
        st = self._stats[name]
        i = entropy_threshold_bin(st["hist"])
        return (i + 0.5) * st["range"] / _NUM_BINS / 127.0


def entropy_threshold_bin(hist, num_levels=_NUM_LEVELS):
    """Clip bin minimising KL(P || Q) as described in the module docstring.  All candidate clip
    bins are evaluated at once (a [candidates, bins] batch) instead of one Python iteration each."""
    hist = hist.double().cpu()
    n = hist.numel()
    cand = torch.arange(num_levels, n + 1)                       # i = number of bins kept
    k = torch.arange(n)
    keep = k[None, :] < cand[:, None]                            # [I, n]
    csum = torch.cumsum(hist, 0)
    tail = csum[-1] - csum[cand - 1]                             # mass beyond the clip bin
    src = hist[None, :] * keep                                   # hist[:i]
    p = src.clone()
    p[torch.arange(cand.numel()), cand - 1] += tail              # outliers folded into the last bin
    # merge the i kept bins into num_levels levels, expand back over the non-empty bins
    idx = torch.ceil((k[None, :].double() + 0.5) * num_levels / cand[:, None].double()).long() - 1
    idx = idx.clamp_(0, num_levels - 1)
    nonzero = (src > 0).double()
    level_sum = torch.zeros(cand.numel(), num_levels, dtype=torch.float64).scatter_add_(1, idx, src)
    level_cnt = torch.zeros(cand.numel(), num_levels, dtype=torch.float64).scatter_add_(1, idx, nonzero)
    q = torch.gather(level_sum / level_cnt.clamp(min=1), 1, idx) * nonzero
    psum, qsum = p.sum(1, keepdim=True), q.sum(1, keepdim=True)
    ok = (psum > 0) & (qsum > 0)
    pn = p / psum.clamp(min=1e-300)
    qn = (q / qsum.clamp(min=1e-300)).clamp(min=1e-12)           # P > 0 with Q == 0: the folded bin only
    term = torch.where(pn > 0, pn * torch.log(pn.clamp(min=1e-300) / qn), torch.zeros_like(pn))
    kl = term.sum(1)
    kl[~ok[:, 0]] = math.inf
    best = int(torch.argmin(kl))                                 # first minimum, like the loop
    return int(cand[best]) - 1 if math.isfinite(float(kl[best])) else n - 1


CALIBRATORS = {"minmax": MinMaxCalibrator, "entropy": EntropyCalibrator, "percentile": PercentileCalibrator,
               "legacy": PercentileCalibrator}


def get_calibrator(calibrator):
    """Factory with the reference's names (det2trt/quantization/calibrator_trt.py:6-16)."""
    assert calibrator in CALIBRATORS, f"calibrator should be in {sorted(CALIBRATORS)}"
    return CALIBRATORS[calibrator]
