"""Readers of the TEXMEX vector-file formats SIFT-1M is distributed in (`sift_base.fvecs`, `sift_query.fvecs`,
`sift_groundtruth.ivecs`; `.bvecs` for the byte variants): every row is `int32 d` followed by d values (f32 / int32 / u8),
little-endian.  The reference converts the same dataset to HDF5 first (py/create_1m_hdf5.py, dataset names `train` / `test` /
`neighbors`); there is no h5py here, so the bench reads the original files directly (bench.py --sift-dir).  No dataset is
available offline: without the directory the bench synthesises SIFT-like rows and says so in its line."""
import os

import numpy as np


def _read_vecs(path, dtype, limit=None):
    item = np.dtype(dtype).itemsize
    with open(path, "rb") as f:
        head = f.read(4)
        if len(head) < 4:
            return np.zeros((0, 0), dtype)
        d = int(np.frombuffer(head, "<i4")[0])
        if d <= 0:
            raise ValueError("%s: bad dimension %d" % (path, d))
        row = 4 + d * item
        size = os.fstat(f.fileno()).st_size
        if size % row:
            raise ValueError("%s: size %d is not a whole number of %d-byte rows" % (path, size, row))
        n = size // row if limit is None else min(size // row, int(limit))
        f.seek(0)
        raw = np.fromfile(f, np.uint8, n * row).reshape(n, row)
    dims = raw[:, :4].copy().view("<i4").reshape(-1)
    if not np.all(dims == d):
        raise ValueError("%s: rows of different dimensions" % path)
    return np.ascontiguousarray(raw[:, 4:]).view(np.dtype(dtype).newbyteorder("<")).reshape(n, d).astype(dtype, copy=False)


def read_fvecs(path, limit=None):
    return _read_vecs(path, np.float32, limit)


def read_ivecs(path, limit=None):
    return _read_vecs(path, np.int32, limit)


def read_bvecs(path, limit=None):
    return _read_vecs(path, np.uint8, limit)


def write_fvecs(path, rows):
    rows = np.ascontiguousarray(rows, np.float32)
    n, d = rows.shape
    out = np.empty((n, 4 + 4 * d), np.uint8)
    out[:, :4] = np.frombuffer(np.int32(d).tobytes(), np.uint8)
    out[:, 4:] = rows.view(np.uint8).reshape(n, 4 * d)
    out.tofile(path)


def write_ivecs(path, rows):
    rows = np.ascontiguousarray(rows, np.int32)
    n, d = rows.shape
    out = np.empty((n, 4 + 4 * d), np.uint8)
    out[:, :4] = np.frombuffer(np.int32(d).tobytes(), np.uint8)
    out[:, 4:] = rows.view(np.uint8).reshape(n, 4 * d)
    out.tofile(path)


def load_sift(directory, n=None, nq=None):
    """(base f32 [n,128], queries f32 [nq,128], ground truth int32 [nq,100] or None) from a SIFT-1M directory
    (sift_base.fvecs / sift_query.fvecs / sift_groundtruth.ivecs; the ANN_SIFT1M tarball's names), or None if absent."""
    bp, qp, gp = (os.path.join(directory, f) for f in ("sift_base.fvecs", "sift_query.fvecs", "sift_groundtruth.ivecs"))
    if not (os.path.exists(bp) and os.path.exists(qp)):
        return None
    base, q = read_fvecs(bp, n), read_fvecs(qp, nq)
    gt = read_ivecs(gp, nq) if os.path.exists(gp) else None
    return base, q, gt
