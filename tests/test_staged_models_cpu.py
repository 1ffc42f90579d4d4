"""Host models of the STAGED kernels' index arithmetic (tools/probes/tsgemm_ares_model.py): every address expression of
tsgemm_s8_ares_kernel / tsgemm_f16_ares_kernel transcribed per thread and executed with numpy must reproduce a @ w.T.
Not a test of the kernels (they have not run on a device yet) -- of the design they implement."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    spec = importlib.util.spec_from_file_location("tsgemm_ares_model", os.path.join(ROOT, "tools", "probes", "tsgemm_ares_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("M,N,K", [(161, 256, 128), (200, 512, 256)])
def test_int8_a_resident_index_arithmetic(model, M, N, K):
    assert model.run(M, N, K, n_blocks=2)


@pytest.mark.parametrize("M,N,K,stages,units", [(200, 256, 128, 3, 3), (230, 512, 256, 2, 5)])
def test_fp16_a_resident_index_arithmetic(model, M, N, K, stages, units):
    assert model.run_f16(M, N, K, stages, units, n_blocks=2)
