"""CPU tests of the host-side tools around the path: the TEXMEX (.fvecs / .ivecs) readers behind `bench.py --sift-dir`, and
bench.py's launcher contract (`--gpus N` never measures fewer ranks than asked)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from muopdb_amd import datasets as DS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fvecs_ivecs_round_trip_and_sift_directory(tmp_path):
    rng = np.random.default_rng(1)
    base = rng.integers(0, 219, (37, 128)).astype(np.float32)
    q = rng.integers(0, 219, (5, 128)).astype(np.float32)
    gt = rng.integers(0, 37, (5, 100)).astype(np.int32)
    DS.write_fvecs(tmp_path / "sift_base.fvecs", base)
    DS.write_fvecs(tmp_path / "sift_query.fvecs", q)
    DS.write_ivecs(tmp_path / "sift_groundtruth.ivecs", gt)
    raw = (tmp_path / "sift_base.fvecs").read_bytes()
    assert len(raw) == 37 * (4 + 128 * 4) and raw[:4] == (128).to_bytes(4, "little")     # int32 d, then d little-endian f32, per row
    assert raw[516:520] == (128).to_bytes(4, "little")
    b2, q2, g2 = DS.load_sift(str(tmp_path))
    assert np.array_equal(b2, base) and np.array_equal(q2, q) and np.array_equal(g2, gt)
    assert DS.load_sift(str(tmp_path), n=10, nq=2)[0].shape == (10, 128)
    assert DS.load_sift(str(tmp_path / "missing")) is None
    (tmp_path / "bad.fvecs").write_bytes(raw[:-3])                                         # truncated file: refused, not misread
    with pytest.raises(ValueError):
        DS.read_fvecs(tmp_path / "bad.fvecs")
    u8 = rng.integers(0, 256, (4, 16)).astype(np.uint8)
    rows = np.concatenate([np.tile(np.frombuffer(np.int32(16).tobytes(), np.uint8), (4, 1)), u8], axis=1)
    rows.tofile(tmp_path / "x.bvecs")
    assert np.array_equal(DS.read_bvecs(tmp_path / "x.bvecs"), u8)


def test_bench_gpus_flag_never_runs_fewer_ranks_silently():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the launcher would really start two ranks")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MDB_BENCH_DEVICE", "MDB_BENCH_BACKEND"):
        env.pop(k, None)
    # no launcher, --gpus 2, fewer than 2 devices: a loud non-zero exit, no JSON line
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--gpus 2" in p.stderr and '"metric"' not in p.stdout
    # a launcher whose world size contradicts the flag is refused as well
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr
