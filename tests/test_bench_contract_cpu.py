"""bench.py pieces that do not need a GPU: the roofline numerator (algorithmic bytes per launch,
SURVEY.md 8d) and the camera partition used for N > 1."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey_8d():
    b = _bench()
    # fp16 base SCA 590.1 MB, base TSA 97.6 MB, base decoder 21.1 MB; INT8 counts 1 byte
    # per element incl. the reference points (bench.py's int8 line), fp32 = 2 x fp16
    assert b.msda_bytes(b.BASE["sca"], 2) == 590054432
    assert round(b.msda_bytes(b.BASE["tsa"], 2) / 1e6, 1) == 97.6
    assert round(b.msda_bytes(b.BASE["dec"], 2) / 1e6, 1) == 21.1
    assert b.msda_bytes(b.BASE["sca"], 4) == 2 * (590054432 - 32) + 32
    # one camera of the SCA call (the unit a rank owns when the 6 cameras are sharded)
    per_cam = (b.msda_bytes(b.BASE["sca"], 2) - 32) // 6
    assert b.msda_bytes(b.BASE["sca"], 2, bs=1) == per_cam + 32
    assert b.HBM_PEAK_GBS == 8000.0


def test_workload_shapes_are_the_base_config():
    b = _bench()
    s = b.BASE["sca"]
    assert s["bs"] == 6 and s["nq"] == 200 * 200 and s["P"] == 8 and s["ppg"] == 4
    assert sum(h * w for h, w in s["levels"]) == 30825
    assert b.BASE["dcn"] == [(23, 256, 58, 100), (3, 512, 29, 50)]     # R101-DCN stages 3 / 4 at 928x1600
    assert b.BASE["enc_layers"] == 6 and b.BASE["dec_layers"] == 6


def test_camera_partition_for_every_gpu_count():
    from bevformer_tensorrt_amd.camera_shard import camera_shards
    for world in (1, 2, 4, 8):
        sh = camera_shards(6, world)
        assert sorted(c for r in sh for c in r) == list(range(6))
        assert max(len(r) for r in sh) == -(-6 // world)
