import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_tensorrt_amd.functions import conv as CV
from bevformer_tensorrt_amd.utils import lib as _lib
h = _lib.load_library()
fn = h.bevops_conv3x3_c64_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
g = torch.Generator().manual_seed(0)
B, H, W = 6, 232, 400
x = torch.randn(B, 64, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().cuda()
wt = CV.pack_taps(w)
b = torch.randn(64, generator=g).half().cuda()
out = torch.empty_like(x)
st = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = fn(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, 1, st.data_ptr(), None)
    torch.cuda.synchronize()
assert rc == 0
s = st.view(256, 8, 8).cpu().double()
tot = (s[:, :, 7] - s[:, :, 6])
mult = ["zero acc", "multiply", "barrier A", "stage outputs", "barrier B", "-"]
move = ["store previous outputs", "issue next loads", "barrier A", "land next pixels", "barrier B", "-"]
print(json.dumps({"kernel_cycles_mean": float(tot.mean()), "tiles_per_block": 2250 / 256,
                  "multiply_waves": {n: float(s[:, :4, k].mean()) for k, n in enumerate(mult[:5])},
                  "mover_waves": {n: float(s[:, 4:, k].mean()) for k, n in enumerate(move[:5])}}))
