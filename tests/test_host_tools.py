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


def _shared_build_worker(rank, world, port, tmp, out):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MDB_BENCH_TMP=tmp)
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    env = object.__new__(bench.Env)
    env.rank, env.world = rank, world
    env.keep_tags, env.kept_builds = {"kept"}, {}
    calls = []

    def build():
        calls.append(rank)
        return dict(index=bytes(range(256)) * 4096, vectors=np.arange(1 << 19, dtype=np.uint8).reshape(-1), codebook=np.arange(24, dtype=np.float32),
                    n=12345, name="c5")

    got, cleanup = env.shared_build("unit", build)
    ok = (calls == ([0] if rank == 0 else [])                                    # ONE builder per job
          and bytes(np.ascontiguousarray(got["index"]).tobytes()) == bytes(range(256)) * 4096
          and np.array_equal(np.asarray(got["vectors"]), np.arange(1 << 19, dtype=np.uint8))
          and np.array_equal(got["codebook"], np.arange(24, dtype=np.float32)) and got["n"] == 12345 and got["name"] == "c5")
    base = os.path.join(tmp, "mdb_bench_%s_unit" % port)
    there = os.path.isdir(base)
    cleanup()
    dist.barrier()
    after = os.path.isdir(base)
    # a tag in keep_tags survives its cleanup, a second shared_build of it maps the same files WITHOUT building, drop_kept_builds removes it
    n0 = len(calls)
    g1, c1 = env.shared_build("kept", build)
    c1()
    kbase = os.path.join(tmp, "mdb_bench_%s_kept" % port)
    ok = ok and os.path.isdir(kbase) and "kept" in env.kept_builds
    g2, c2 = env.shared_build("kept", build)
    ok = ok and len(calls) == n0 + (1 if rank == 0 else 0) and np.array_equal(np.asarray(g2["vectors"]), np.asarray(g1["vectors"])) and g2["n"] == 12345
    del g1, g2
    c2()
    env.drop_kept_builds()
    dist.barrier()
    ok = ok and not os.path.isdir(kbase) and env.kept_builds == {}
    out.put((rank, ok, there, after))
    dist.destroy_process_group()


def test_bench_shared_build_one_builder_per_job(tmp_path):
    """bench.py's Env.shared_build under world_size-2 gloo: rank 0 builds, both ranks read the same bytes (memory-mapped files under
    MDB_BENCH_TMP), the small values travel by pickle, cleanup removes the directory after every rank has loaded."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_shared_build_worker, args=(r, 2, port, str(tmp_path), out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok and there and not after for _, ok, there, after in res), res


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, root


def test_bench_final_line_is_bounded_and_parses():
    """the driver parses ONE stdout line from a bounded tail — round 4's 24.8 KB line did not parse.  A canned FULL
    record of that run (profiles/r04_bench_all.json, 10 workloads with dispersion blocks, prose and sweeps) must compact to < 8 KB,
    round-trip through json and still carry the contract's fields, the roofline and the cpu_baseline at the top level."""
    import json
    bench, root = _load_bench()
    with open(os.path.join(root, "profiles", "r04_bench_all.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000   # the canned record IS the oversized one
    # a worst case on top: twice the workloads, an error entry with a long message
    for name in list(full["workloads"]):
        full["workloads"][name + "_again"] = full["workloads"][name]
    full["workloads"]["broken"] = {"error": "RuntimeError: " + "x" * 5000}
    text = bench.compact_line(full)
    assert "\n" not in text and len(text) <= bench.LINE_LIMIT < 8192
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "workloads"):
        assert key in line, key
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "bytes_per_launch", "kernel_ms"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "all_cores_value", "cpu_model", "ids_match_gpu"}
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) / full["roofline"]["frac"] < 1e-3
    assert len(line["config"]["workload"]) < 80 and "model" not in line["config"]
    w = line["workloads"]["ivfpq_c3"]
    assert abs(w["value"] - full["workloads"]["ivfpq_c3"]["value"]) / w["value"] < 1e-3 and "frac" in w and "cpu1" in w
    # the plain r04 record (10 workloads) keeps every entry
    with open(os.path.join(root, "profiles", "r04_bench_all.json")) as f:
        plain = json.load(f)
    line = json.loads(bench.compact_line(plain))
    assert set(line["workloads"]) == set(plain["workloads"]) and "workloads_truncated" not in line
