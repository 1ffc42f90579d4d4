"""linear_bias_act -- the dense layers that wrap the sampler with their whole
epilogue in the GEMM: out = act(x @ weight.T + bias + residual) as one `bevops_linear_bias_act`
call (hipBLASLt MFMA kernel, shift + identity + ReLU in its epilogue).  Not a reference plugin:
the reference leaves these layers to cuBLAS / TensorRT."""
import torch

from ..utils import lib as _lib
from ..utils import workspace as _ws

def _workspace(device, nbytes, stream):
    """hipBLASLt workspace per (device, stream): two streams never share one (utils/workspace.py)."""
    return _ws.lend("linear", nbytes, device, stream)


_TUNED = set()   # problems whose algorithm has been chosen by bevops_linear_tune in this process


def _tune_once(handle, key, args, x2, out_shape):
    """The library never synchronises inside the operator; algorithm selection by measurement is
    its own BLOCKING entry (bevops_linear_tune), called here once per problem, outside stream
    capture, with a scratch output (BEVOPS_LINEAR_TUNE=0 switches it off)."""
    import os
    _TUNED.add(key)
    if os.environ.get("BEVOPS_LINEAR_TUNE", "1") == "0":
        return
    scratch = torch.empty(out_shape, dtype=x2.dtype, device=x2.device)
    a = list(args)
    a[5] = scratch.data_ptr()
    handle.bevops_linear_tune(*a)       # status ignored: the heuristic's choice stays on failure


def linear_bias_act(x, weight, bias=None, residual=None, relu=False, out=None):
    """x [..., K] fp16 (contiguous rows), weight [N, K], bias [N] or None, residual [..., N] or None
    -> [..., N].  `out` may be given (and may be `residual` itself).  Raises BevopsError with status
    NOT_SUPPORTED when hipBLASLt has no algorithm for the shape (callers fall back to
    mm + bias_act_nhwc_)."""
    assert x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match x [..., {K}]")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    weight = weight.contiguous()
    M = x2.shape[0]
    if residual is not None:
        if residual.dtype != x.dtype or residual.numel() != M * N:
            raise ValueError("residual must be fp16 with M*N elements")
        r2 = residual.reshape(M, N)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    else:
        r2 = None
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == M * N and out.dtype == x.dtype
    if M == 0:
        return out.view(*x.shape[:-1], N)
    handle = _lib.load_library()
    import os
    nbytes = handle.bevops_linear_workspace_size() if os.environ.get("BEVOPS_LINEAR_WS", "1") != "0" else 0
    stream = _lib.current_stream_ptr(x.device)
    ws = _workspace(x.device, nbytes, stream) if nbytes else None
    args = (_lib.F16, x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
            r2.data_ptr() if r2 is not None else None, out.data_ptr(), M, N, K, int(bool(relu)),
            ws.data_ptr() if ws is not None else None, nbytes, stream)
    with torch.cuda.device(x.device):
        key = (str(x.device), M, N, K, bool(relu), bias is not None, r2 is not None)
        if key not in _TUNED and not torch.cuda.is_current_stream_capturing():
            _tune_once(handle, key, args, x2, (M, N))
        st = handle.bevops_linear_bias_act(*args)
    _lib.check(st, "bevops_linear_bias_act")
    return out.view(*x.shape[:-1], N)


def layer_norm(x, weight=None, bias=None, eps=1e-5, out=None):
    """torch.nn.functional.layer_norm over the last dimension (fp16, C in {64,128,256,512}) as one
    streaming pass (bevops_layer_norm).  Raises BevopsError (status 3) for other widths."""
    assert x.is_cuda and x.dtype == torch.float16
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    if out is None:
        out = torch.empty_like(x2)
    else:
        assert out.is_contiguous() and out.numel() == x2.numel() and out.dtype == x.dtype
    w = weight.to(torch.float16).contiguous() if weight is not None else None
    b = bias.to(torch.float16).contiguous() if bias is not None else None
    if x2.shape[0] == 0:
        return out.view(x.shape)
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_layer_norm(_lib.F16, x2.data_ptr(), w.data_ptr() if w is not None else None,
                                      b.data_ptr() if b is not None else None, out.data_ptr(), x2.shape[0], C,
                                      float(eps), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_layer_norm")
    return out.view(x.shape)


def quantize_rows(x, scale, out=None):
    """fp16 tensor -> int8 with one per-tensor scale: clamp(rne(x / scale), -127, 127) (bevops_quantize_rows)."""
    assert x.is_cuda and x.dtype == torch.float16
    x = x.contiguous()
    if x.numel() % 8:
        raise ValueError("element count must be a multiple of 8")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    if x.numel() == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_quantize_rows(_lib.F16, x.data_ptr(), out.data_ptr(), x.numel(), float(scale),
                                         _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_quantize_rows")
    return out


def dequantize_rows(q, scale):
    """int8 tensor -> fp16, q * scale with the product in fp32 and one rounding (bevops_dequantize_rows)."""
    assert q.is_cuda and q.dtype == torch.int8
    q = q.contiguous()
    if q.numel() % 8:
        raise ValueError("element count must be a multiple of 8")
    out = torch.empty(q.shape, dtype=torch.float16, device=q.device)
    if q.numel() == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(q.device):
        st = handle.bevops_dequantize_rows(_lib.F16, q.data_ptr(), out.data_ptr(), q.numel(), float(scale),
                                           _lib.current_stream_ptr(q.device))
    _lib.check(st, "bevops_dequantize_rows")
    return out


def linear_int8(a_q, scale_a, w_q, scale_w, bias=None, residual=None, relu=False, out_dtype=torch.float16,
                scale_out=1.0):
    """INT8 GEMM with de-quantising epilogue: a_q [..., K] int8 (bevops_linear_int8) -- or the fp16 activation
    itself, quantised with scale_a inside the GEMM's operand load (bevops_linear_int8_fused: no quantise pass)
    --, w_q [N, K] int8, scale_w a float (per tensor) or an fp32 [N] tensor (per output channel), bias fp32
    [N], residual fp16 [..., N] -> fp16 (or int8 requantised with scale_out)."""
    assert a_q.is_cuda and a_q.dtype in (torch.int8, torch.float16) and w_q.dtype == torch.int8
    fused = a_q.dtype == torch.float16
    K, N = a_q.shape[-1], w_q.shape[0]
    a2 = a_q.reshape(-1, K).contiguous()
    w_q = w_q.contiguous()
    M = a2.shape[0]
    per_channel = torch.is_tensor(scale_w)
    ws = scale_w.float().contiguous() if per_channel else None
    b = bias.float().contiguous() if bias is not None else None
    r = residual.reshape(M, N).contiguous() if residual is not None else None
    out = torch.empty((M, N), dtype=out_dtype, device=a_q.device)
    if M == 0:
        return out.view(*a_q.shape[:-1], N)
    handle = _lib.load_library()
    with torch.cuda.device(a_q.device):
        st = (handle.bevops_linear_int8_fused if fused else handle.bevops_linear_int8)(
            a2.data_ptr(), float(scale_a), w_q.data_ptr(), ws.data_ptr() if per_channel else None,
            1.0 if per_channel else float(scale_w), b.data_ptr() if b is not None else None,
            r.data_ptr() if r is not None else None, _lib.torch_dtype_code(out), out.data_ptr(), float(scale_out),
            M, N, K, int(bool(relu)), _lib.current_stream_ptr(a_q.device))
    _lib.check(st, "bevops_linear_int8_fused" if fused else "bevops_linear_int8")
    return out.view(*a_q.shape[:-1], N)


def tsgemm(x, weight, bias=None, residual=None, relu=False, out=None):
    """act(x @ weight.T + bias + residual) on the hand-written tall-skinny MFMA GEMM (bevops_tsgemm_f16):
    x [..., K] fp16 contiguous rows, weight [N, K], residual / out [..., N].  Raises BevopsError with status
    NOT_SUPPORTED outside its domain (K % 64, N % 256)."""
    assert x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16
    K, N = x.shape[-1], weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match x [..., {K}]")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    weight = weight.contiguous()
    M = x2.shape[0]
    r2 = None
    if residual is not None:
        r2 = residual.reshape(M, N)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if M == 0:
        return out.view(*x.shape[:-1], N)
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_tsgemm_f16(x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      r2.data_ptr() if r2 is not None else None, out.data_ptr(), M, N, K,
                                      int(bool(relu)), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_tsgemm_f16")
    return out.view(*x.shape[:-1], N)


def tsgemm_ln(x, weight, bias, residual, ln_weight, ln_bias, eps=1e-5):
    """layer_norm(x @ weight.T + bias + residual) * ln_weight + ln_bias in ONE launch (bevops_tsgemm_f16_ln): the dense
    layer that ends an attention / FFN block of the encoder or decoder together with the block's norm
    (modules/encoder.py:586-636).  x [..., K] fp16, weight [256, K], residual [..., 256] or None.  Equal to
    layer_norm(tsgemm(...)) up to the last bit of the normalisation (same binary16 sums, fp32 statistics).  Raises
    BevopsError (NOT_SUPPORTED) unless N == 256 and K % 64 == 0."""
    assert x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16
    K, N = x.shape[-1], weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match x [..., {K}]")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    weight = weight.contiguous()
    M = x2.shape[0]
    r2 = None
    if residual is not None:
        r2 = residual.reshape(M, N)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    bias = None if bias is None else bias.to(torch.float16).contiguous()
    g, b = ln_weight.to(torch.float16).contiguous(), ln_bias.to(torch.float16).contiguous()
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if M == 0:
        return out.view(*x.shape[:-1], N)
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_tsgemm_f16_ln(x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                         r2.data_ptr() if r2 is not None else None, g.data_ptr(), b.data_ptr(), float(eps),
                                         out.data_ptr(), M, N, K, _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_tsgemm_f16_ln")
    return out.view(*x.shape[:-1], N)


def tile_gemm(x, weight, bias=None, residual=None, relu=False, out=None):
    """act(x @ weight.T + bias + residual) on the tiled MFMA GEMM (bevops_tile_gemm_f16, csrc/tile_gemm.hip:
    128 x 128 tiles, three blocks per CU): x [..., K] fp16 contiguous rows, weight [N, K], bias [N] fp16,
    residual / out [..., N].  K % 8 == 0."""
    assert x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16
    K, N = x.shape[-1], weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match x [..., {K}]")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    weight = weight.contiguous()
    M = x2.shape[0]
    r2 = None
    if residual is not None:
        if residual.dtype != x.dtype or residual.numel() != M * N:
            raise ValueError("residual must be fp16 with M*N elements")
        r2 = residual.reshape(M, N)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == M * N and out.dtype == x.dtype
    if M == 0:
        return out.view(*x.shape[:-1], N)
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_tile_gemm_f16(x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                         r2.data_ptr() if r2 is not None else None, out.data_ptr(), M, N, K,
                                         int(bool(relu)), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_tile_gemm_f16")
    return out.view(*x.shape[:-1], N)


def small_gemm(x, weight, bias=None, residual=None, relu=False, out=None):
    """act(x @ weight.T + bias + residual) for layers with FEW rows (the decoder's 900 queries) on the
    no-pipeline MFMA GEMM (bevops_small_gemm_f16, csrc/small_gemm.hip: 32 x 64 tiles, split-K inside the block, every
    operand fragment requested before the first matrix instruction -- one memory round trip per launch instead of a
    chain of dependent k-steps).  x [..., K] fp16, weight [N, K]; K % 64 == 0, K <= 1024; offered up to 8192 rows."""
    assert x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16
    K, N = x.shape[-1], weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match x [..., {K}]")
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M > 8192:
        raise _lib.BevopsError("small_gemm: more than 8192 rows (the tiled GEMMs' domain)", _lib.NOT_SUPPORTED)
    weight = weight.contiguous()
    r2 = None
    if residual is not None:
        if residual.dtype != x.dtype or residual.numel() != M * N:
            raise ValueError("residual must be fp16 with M*N elements")
        r2 = residual.reshape(M, N)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == M * N and out.dtype == x.dtype
    if M == 0:
        return out.view(*x.shape[:-1], N)
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_small_gemm_f16(x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          r2.data_ptr() if r2 is not None else None, out.data_ptr(), M, N, K,
                                          int(bool(relu)), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_small_gemm_f16")
    return out.view(*x.shape[:-1], N)


# ---- measured choice between the dense-layer implementations -------------------------------------------------
def _torch_dense(x, weight, bias, residual, relu):
    if residual is not None or bias is None:
        raise _lib.BevopsError("torch addmm path: bias required, no identity term", _lib.NOT_SUPPORTED)
    x2 = x.reshape(-1, x.shape[-1])
    y = torch._addmm_activation(bias, x2, weight.t()) if relu else torch.addmm(bias, x2, weight.t())
    return y.view(*x.shape[:-1], weight.shape[0])


_DENSE = {"tsgemm": tsgemm, "tile": tile_gemm, "small": small_gemm, "blaslt": linear_bias_act, "torch": _torch_dense}
_DENSE_CHOICE = {}     # problem -> name of the fastest implementation measured in this process
DENSE_LOG = []         # (problem, {name: us}) of every measurement, for tools / profiles
DENSE_MISSES = []      # problems dense_auto met that dispatch_gfx950.json does not list (measured or defaulted instead)
# Reproducible mode: no per-process timing, the choice is a function of the problem alone and falls on the two
# hand-written kernels (fixed summation order, no library heuristic).  The camera-sharded frame loop switches it on:
# ranks that each measured their own winner would evaluate the REPLICATED layers (TSA, FFN, decoder) in different
# summation orders and drift apart bitwise.  BEVOPS_DENSE_TUNE=0 is the same switch from the environment.
class _PerThreadFlag:
    """`flag["enabled"]` with one value per THREAD (two frame runners on two threads -- a sharded and a plain one --
    must not see each other's setting; advisor, round 4).  Dict-style access, default False."""

    def __init__(self):
        import threading
        self._tls = threading.local()

    def __getitem__(self, key):
        assert key == "enabled"
        return getattr(self._tls, "enabled", False)

    def __setitem__(self, key, value):
        assert key == "enabled"
        self._tls.enabled = bool(value)


DETERMINISTIC = _PerThreadFlag()
# The hand-written kernels only, the FASTEST of them per problem by the shipped table's own measurements (the rule of
# DETERMINISTIC for a problem the table has not seen): what the model runs behind its backbone (bevformer.py:
# _OWN_ENCODER) -- one kernel per layer, summation order a function of the block index, same choice in every process.
OWN_KERNELS = _PerThreadFlag()


# Shipped choices (bevformer_tensorrt_amd/dispatch_gfx950.json, written by tools/dump_dispatch.py from one MI355X's
# measurements): a problem found there takes its recorded winner WITHOUT a measurement, so the kernel that runs is the
# same on every box and in every process; only problems the table has never seen are timed (BEVOPS_DENSE_TUNE=1: time
# everything, i.e. regenerate; =0: no table, no timing -- the reproducible defaults).
_TABLE = {"loaded": False, "dense": {}, "conv": {}, "measured_dense": {}}


def _table():
    import json
    import os
    if not _TABLE["loaded"]:
        _TABLE["loaded"] = True
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dispatch_gfx950.json")
        if os.path.exists(path) and os.environ.get("BEVOPS_DENSE_TUNE", "") not in ("0", "1"):
            try:
                t = json.load(open(path))
                _TABLE["dense"], _TABLE["conv"] = t.get("dense", {}), t.get("conv", {})
                _TABLE["measured_dense"] = t.get("measured_us", {}).get("dense", {})
            except (OSError, ValueError):
                pass
    return _TABLE


def _problem(key):
    return ",".join(str(int(v)) if isinstance(v, bool) else str(v) for v in key[1:])   # without the device name


def _small_pays(M, N, K):
    """The no-pipeline few-row GEMM is for problems of a few tiles (the decoder's 900 object queries); the shipped
    table's own measurements show it 2-3 x slower than the tiled kernels once M x N grows (2 250 x 2 048 x 512: 33.8
    against 18.0 us)."""
    return K % 64 == 0 and K <= 1024 and M * N <= 1024 * 512


def _dense_deterministic(N, K, M=1 << 30):
    if _small_pays(M, N, K):
        return "small"
    return "tsgemm" if (N % 256 == 0 and K % 64 == 0 and K >= 256) else "tile"


def _dense_own(key, N, K, M):
    times = _table().get("measured_dense", {}).get(_problem(key), {})
    own = {k: v for k, v in times.items() if k in ("tile", "tsgemm", "small") and k in _DENSE}
    return min(own, key=own.get) if own else _dense_deterministic(N, K, M)


def _dense_default(N, K, has_res):
    """Choice without a measurement (inside stream capture before the problem was seen, or BEVOPS_DENSE_TUNE=0)."""
    if N == 256 and K % 64 == 0 and K >= 256:
        return "tsgemm"
    return "blaslt"


def dense_auto(x, weight, bias=None, residual=None, relu=False):
    """act(x @ weight.T + bias + residual), fp16, on whichever of the implementations is fastest for the
    problem on THIS device: the tall-skinny persistent GEMM (tsgemm), the tiled GEMM (tile_gemm), the hipBLASLt
    entry with the fused epilogue (linear_bias_act) or the framework's addmm.  Measured once per
    (M, N, K, epilogue) outside stream capture -- BLOCKING, like bevops_linear_tune: a few launches of each
    candidate into scratch outputs between device synchronisations -- then cached for the process.  All
    candidates accumulate in fp32 and round once; they differ in summation order only."""
    import os
    K, N = x.shape[-1], weight.shape[0]
    M = x.numel() // K
    if M == 0:          # a rank of the camera-sharded path that owns no camera
        return x.new_empty((*x.shape[:-1], N))
    key = (str(x.device), M, N, K, bool(relu), bias is not None, residual is not None)
    if DETERMINISTIC["enabled"] and K % 8 == 0:
        name = _dense_deterministic(N, K, M)
    elif OWN_KERNELS["enabled"] and K % 8 == 0:
        name = _dense_own(key, N, K, M)
    else:
        name = _DENSE_CHOICE.get(key)
    if name is None:
        name = _table()["dense"].get(_problem(key))
        if name is not None and name in _DENSE:
            _DENSE_CHOICE[key] = name
        else:
            name = None
            if _problem(key) not in DENSE_MISSES:
                DENSE_MISSES.append(_problem(key))     # a problem the shipped table has never seen (bench.py reports them)
    if name is None:
        if torch.cuda.is_current_stream_capturing() or os.environ.get("BEVOPS_DENSE_TUNE", "1") == "0" or M < 64:
            name = "small" if _small_pays(M, N, K) else _dense_default(N, K, residual is not None)
        else:
            name = _DENSE_CHOICE[key] = _dense_measure(key, x, weight, bias, residual, relu)
    try:
        return _DENSE[name](x, weight, bias, residual, relu)
    except _lib.BevopsError as exc:
        if exc.status != _lib.NOT_SUPPORTED or name == "blaslt":
            raise
        return linear_bias_act(x, weight, bias, residual, relu)


def graph_time_us(fn, iters=8, rounds=3):
    """Microseconds per call of `fn` on the device, measured under HIP-graph replay: `iters` calls are captured once
    and the replay is timed between two events (best of `rounds`).  Launching the same calls eagerly measures the HOST
    for anything shorter than the ~12 us a Python operator wrapper takes per call -- the decoder's few-row layers all
    looked alike (12-13 us) that way, whatever the kernel did."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    best = float("inf")
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


def _dense_measure(key, x, weight, bias, residual, relu, rounds=3, iters=8):
    times = {}
    with torch.cuda.device(x.device):
        for name, fn in _DENSE.items():
            try:
                for _ in range(2):
                    fn(x, weight, bias, residual, relu)
                times[name] = float("inf")
            except _lib.BevopsError as exc:   # outside the candidate's domain / no library algorithm -- and only that:
                if exc.status != _lib.NOT_SUPPORTED:      # a launch failure must not be mistaken for "not applicable"
                    raise
                continue
        torch.cuda.synchronize()
        for name in list(times):
            times[name] = graph_time_us(lambda: _DENSE[name](x, weight, bias, residual, relu), iters, rounds)
    DENSE_LOG.append((key, {k: round(v, 1) for k, v in times.items()}))
    if not times:
        return "blaslt"
    return min(times, key=times.get)


def tsa_split(both, heads, points):
    """The stacked sampling_offsets | attention_weights projection of temporal self-attention, [nq, heads * 2 * points * 3]
    fp16 with the reference's column order ([heads][queue][points][xy] | [heads][queue][points]), as the queue-major
    operands of the MSDA call: (offsets [2, nq, heads, points * 2], weights [2, nq, heads, points]) in ONE pass
    (bevops_tsa_split) instead of two permute-copies."""
    assert both.is_cuda and both.dtype == torch.float16 and both.dim() == 2 and both.is_contiguous()
    nq = both.shape[0]
    assert both.shape[1] == heads * 2 * points * 3
    off = torch.empty((2, nq, heads, points * 2), dtype=both.dtype, device=both.device)
    w = torch.empty((2, nq, heads, points), dtype=both.dtype, device=both.device)
    handle = _lib.load_library()
    with torch.cuda.device(both.device):
        st = handle.bevops_tsa_split(_lib.F16, both.data_ptr(), off.data_ptr(), w.data_ptr(), nq, heads, points,
                                     _lib.current_stream_ptr(both.device))
    _lib.check(st, "bevops_tsa_split")
    return off, w


def queue_mean2(x):
    """torch.mean(x, dim=0, keepdim=True) for x [2, ...] fp16 as one streaming pass (bevops_queue_mean2)."""
    assert x.is_cuda and x.dtype == torch.float16 and x.shape[0] == 2 and x.is_contiguous()
    out = torch.empty((1,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if out.numel() == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_queue_mean2(_lib.F16, x.data_ptr(), out.data_ptr(), out.numel(), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_queue_mean2")
    return out
