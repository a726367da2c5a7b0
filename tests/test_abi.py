"""CPU-side checks of the drop-in boundary: libmuopdb_hip.so loads and exports every entry point
that include/muopdb_hip.h declares (no compute calls: there is no GPU here), the Python binding lists
the same symbols, and the product fails LOUDLY without a device (no CPU fallback)."""
import os
import re

import pytest

from muopdb_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "muopdb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    decl = declared_symbols()
    assert len(decl) >= 40
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, "libmuopdb_hip.so does not export: %s" % missing
    assert sorted(L.EXPORTED_SYMBOLS) == decl, "muopdb_amd.lib.EXPORTED_SYMBOLS is out of sync with the header"
    assert b"gfx950" in lib.mdb_version()


def test_struct_layouts_match_the_header():
    import ctypes as C
    assert C.sizeof(L.U128) == 16
    assert C.sizeof(L.UserIndexInfoC) == 112           # rs/index/src/multi_spann/user_index_info.rs:26-42
    assert C.sizeof(L.Stats) == 32
    assert L.SearchParamsC.top_k.offset == 0 and L.SearchParamsC.centroid_distance_ratio.offset == 24


def test_size_helpers_need_no_device():
    """the block / row size helpers of the sharded step are pure arithmetic (a host sizes its all-gather buffers before any device call):
    mdb_points_block_bytes, mdb_shard_block_bytes, mdb_spann_probe_row_words (2 + num_explored_centroids, or top_k when unset, at least 1:
    rs/index/src/spann/index.rs:219-222)"""
    import ctypes as C
    lib = L.load()
    assert lib.mdb_shard_block_bytes(C.c_size_t(3), C.c_size_t(10)) == (3 * 10 * 20 + 3 * 4 + 15) // 16 * 16
    assert lib.mdb_points_block_bytes(C.c_size_t(7), C.c_size_t(5)) % 16 == 0 and lib.mdb_points_block_bytes(C.c_size_t(7), C.c_size_t(5)) >= 7 * (8 * 5 + 5)
    p = L.SearchParamsC()
    for top_k, ne, want in ((10, -1, 12), (10, 16, 18), (3, 0, 3), (0, -1, 3), (1, 1, 3)):
        p.top_k, p.ef_construction, p.num_explored_centroids, p.centroid_distance_ratio = top_k, 40, ne, 0.1
        assert lib.mdb_spann_probe_row_words(C.byref(p)) == want, (top_k, ne)
    assert lib.mdb_spann_probe_row_words(None) == 0


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(L.MuopdbError) as e:
        L.Context(0)
    assert e.value.status == 4  # MDB_ERR_HIP


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under muopdb_amd/ may reference it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "muopdb_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "libmuopdb_oracle" not in text, f
