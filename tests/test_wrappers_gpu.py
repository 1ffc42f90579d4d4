"""GPU half of tests/test_wrappers_cpu.py: the re-hosted L3 callers with the HIP operators
(through the C ABI) against the outputs of the reference's own `forward_trt` methods
(tests/golden/wrappers.npz).  fp32: max abs <= 2e-5 (the MSDA kernel's fp32 bar on top of the
hipBLASLt GEMMs); fp16: <= 1e-2 relative to the output scale, north_star's fp16 tolerance."""
import numpy as np
import pytest
import torch

from test_wrappers_cpu import run_dec, run_sca, run_tsa

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("run", [run_sca, run_tsa, run_dec], ids=["sca", "tsa", "decoder_attn"])
def test_rehosted_wrapper_with_hip_ops_equals_reference_forward_trt(run):
    import bevformer_tensorrt_amd.functions as hip_ops
    torch.backends.cuda.matmul.allow_tf32 = False
    got, want = run(hip_ops, "cuda")
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 5e-5, np.abs(got - want).max()


def test_decoder_reference_point_loop_on_gpu():
    """Same loop as the CPU test, on the device: log / sigmoid come from the device's math library
    there, so the bar is 1 ulp-ish (1e-6 on values in (0, 1)), not bit equality."""
    from conftest import golden
    from bevformer_tensorrt_amd import geometry as G
    g = golden("wrappers")
    ref = torch.from_numpy(g["loop.ref0"]).cuda()
    out = torch.from_numpy(g["loop.query"]).cuda()
    for i in range(3):
        out = out + torch.from_numpy(g["loop.steps"][i]).cuda()
        tmp = torch.nn.functional.linear(out, torch.from_numpy(g[f"loop.reg{i}.weight"]).cuda(),
                                         torch.from_numpy(g[f"loop.reg{i}.bias"]).cuda()).view(1, -1, 10)
        ref = G.refine_reference_points(tmp, ref)
        assert np.abs(ref.cpu().numpy() - g["loop.inter_ref"][i]).max() <= 2e-5
