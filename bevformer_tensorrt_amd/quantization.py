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
    """Clip bin minimising KL(P || Q) as described in the module docstring."""
    hist = hist.double()
    n = hist.numel()
    best, best_i = math.inf, n - 1
    total_tail = torch.flip(torch.cumsum(torch.flip(hist, [0]), 0), [0])  # tail[i] = sum hist[i:]
    for i in range(num_levels, n + 1):
        p = hist[:i].clone()
        if i < n:
            p[i - 1] += total_tail[i]
        psum = p.sum()
        if psum <= 0:
            continue
        # merge the i bins into num_levels quantisation levels, expand over non-empty bins
        edges = torch.linspace(0, i, num_levels + 1)
        idx = torch.bucketize(torch.arange(i, dtype=torch.float32) + 0.5, edges[1:-1])
        src = hist[:i]
        level_sum = torch.zeros(num_levels, dtype=torch.float64).index_add_(0, idx, src)
        nonzero = (src > 0).double()
        level_cnt = torch.zeros(num_levels, dtype=torch.float64).index_add_(0, idx, nonzero)
        q = torch.where(nonzero > 0, (level_sum / level_cnt.clamp(min=1))[idx], torch.zeros_like(src))
        qsum = q.sum()
        if qsum <= 0:
            continue
        pn, qn = p / psum, q / qsum
        mask = pn > 0
        # bins where P > 0 but Q == 0 can only be the folded-outlier bin: penalise with a tiny Q
        kl = float((pn[mask] * torch.log(pn[mask] / qn[mask].clamp(min=1e-12))).sum())
        if kl < best:
            best, best_i = kl, i - 1
    return best_i


CALIBRATORS = {"minmax": MinMaxCalibrator, "entropy": EntropyCalibrator, "percentile": PercentileCalibrator,
               "legacy": PercentileCalibrator}


def get_calibrator(calibrator):
    """Factory with the reference's names (det2trt/quantization/calibrator_trt.py:6-16)."""
    assert calibrator in CALIBRATORS, f"calibrator should be in {sorted(CALIBRATORS)}"
    return CALIBRATORS[calibrator]
