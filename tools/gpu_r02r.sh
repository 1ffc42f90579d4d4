#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02r; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mdconv_gpu.py -q -k "lds_dma" 2>&1 | tail -60 ) > $OUT/pytest.log
python - > $OUT/diag.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd.utils import load_library
lib = load_library()
g = torch.Generator().manual_seed(7)
B, Cin, Cout, H, W = 6, 256, 256, 58, 100
x = torch.randint(-128, 128, (B, Cin, H, W), generator=g, dtype=torch.int8).cuda()
off = torch.randint(-127, 128, (B, 18, H, W), generator=g, dtype=torch.int8).cuda()
mask = torch.randint(-8, 128, (B, 9, H, W), generator=g, dtype=torch.int8).cuda()
w = torch.randint(-127, 128, (Cout, Cin, 3, 3), generator=g, dtype=torch.int8).cuda()
b = torch.randn(Cout, generator=g).cuda()
args = (x, off, mask, w, b, 0.02, 0.03, 1 / 127, 0.004, 0.6, 1, 1, 1, 1, 1)
lib.bevops_mdconv_set_variant(6); ref = bev.modulated_deformable_conv2d_int8(*args)
for v in (0, 36, 4, 0, 36):
    lib.bevops_mdconv_set_variant(v)
    outs = [bev.modulated_deformable_conv2d_int8(*args) for _ in range(3)]
    torch.cuda.synchronize()
    d = [(o != ref) for o in outs]
    print("variant", v, "mismatch counts", [int(m.sum()) for m in d], "max abs", [int((o.int() - ref.int()).abs().max()) for o in outs])
    m = d[0]
    if m.any():
        idx = m.nonzero()[:8].tolist()
        print("   first", idx, [ (int(outs[0][tuple(i)]), int(ref[tuple(i)])) for i in idx])
        print("   by channel-quarter", [int(m[:, c::4].sum()) for c in range(4)], "by pixel%128 <64:", int(m.flatten(2)[:, :, :].sum()))
lib.bevops_mdconv_set_variant(0)
PY
tail -30 $OUT/pytest.log; cat $OUT/diag.txt
