"""Host-side logic of the operator layer that needs no GPU: dispatch defaults."""
def test_dense_dispatch_default_without_measurement():
    """functions/linear.py: the choice made when a problem cannot be measured (inside stream capture, or
    BEVOPS_DENSE_TUNE=0): the persistent tall-skinny GEMM for 256-column layers with K >= 256, the library entry
    otherwise; the candidate table names every implementation the measured dispatch picks from."""
    from bevformer_tensorrt_amd.functions import linear as L
    assert L._dense_default(256, 256, True) == "tsgemm"
    assert L._dense_default(256, 1024, False) == "tsgemm"
    assert L._dense_default(256, 64, False) == "blaslt"        # short K: one step per tile, the persistent kernel loses
    assert L._dense_default(512, 256, False) == "blaslt"
    assert L._dense_default(256, 200, False) == "blaslt"       # outside tsgemm's K % 64 domain
    assert set(L._DENSE) == {"tsgemm", "tile", "small", "blaslt", "torch"}
    # reproducible mode (camera-sharded runs, BEVOPS_DENSE_TUNE=0 semantics): a function of the problem alone, on the
    # hand-written kernels only; few rows -> the no-pipeline GEMM
    assert L._dense_deterministic(256, 256, 40000) == "tsgemm"
    assert L._dense_deterministic(192, 256, 40000) == "tile"
    assert L._dense_deterministic(256, 256, 900) == "small"
    assert L._dense_deterministic(256, 200, 900) == "tile"      # outside small_gemm's K % 64 domain
    # shipped table: keys are the problem without the device name
    assert L._problem(("cuda:0", 34800, 256, 1024, True, True, False)) == "34800,256,1024,1,1,0"
