"""Minimal single-file NIfTI-1 (.nii / .nii.gz) reader and writer in numpy.

The reference reads and writes its edited label maps and reconstructions through nibabel (run_recon.py:83-96,
utils/__init__.py:221-228); nibabel is not part of this build, and the subset it uses is small: one array, an affine,
little-endian, no extensions.  Layout per the NIfTI-1 standard (348-byte header, data at vox_offset 352, Fortran order).
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def save(array, path, affine=None):
    """Write `array` (<= 7 dims, any dtype in the table) with `affine` (4x4, default identity) as sform."""
    a = np.asarray(array)
    code = _CODES.get(a.dtype.str[1:])
    if code is None:
        raise ValueError("nifti.save: unsupported dtype %s" % a.dtype)
    if a.ndim < 1 or a.ndim > 7:
        raise ValueError("nifti.save: 1..7 dimensions")
    aff = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    dim = [a.ndim] + list(a.shape) + [1] * (7 - a.ndim)
    pixdim = [1.0] + [float(np.linalg.norm(aff[:3, i])) if i < 3 else 1.0 for i in range(7)]
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<h", hdr, 70, code)
    struct.pack_into("<h", hdr, 72, a.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<f", hdr, 108, 352.0)            # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)              # scl_slope
    struct.pack_into("<f", hdr, 116, 0.0)              # scl_inter
    struct.pack_into("<h", hdr, 252, 0)                # qform_code
    struct.pack_into("<h", hdr, 254, 2)                # sform_code: aligned
    struct.pack_into("<4f", hdr, 280, *aff[0])
    struct.pack_into("<4f", hdr, 296, *aff[1])
    struct.pack_into("<4f", hdr, 312, *aff[2])
    hdr[344:348] = b"n+1\x00"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\x00\x00\x00\x00")                   # no extensions
        f.write(np.asfortranarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="F"))


def load(path):
    """-> (array as stored, scaled by scl_slope / scl_inter when set -> float64 like nibabel's get_fdata; affine 4x4)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError("nifti.load: file too short")
    end = "<"
    if struct.unpack_from("<i", raw, 0)[0] != 348:
        if struct.unpack_from(">i", raw, 0)[0] != 348:
            raise ValueError("nifti.load: not a NIfTI-1 file")
        end = ">"
    if raw[344:347] != b"n+1":
        raise ValueError("nifti.load: only single-file NIfTI-1 (magic n+1) is supported")
    dim = struct.unpack_from(end + "8h", raw, 40)
    code = struct.unpack_from(end + "h", raw, 70)[0]
    if code not in _DTYPES:
        raise ValueError("nifti.load: unsupported datatype code %d" % code)
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    off = int(struct.unpack_from(end + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(end + "2f", raw, 112)
    dt = np.dtype(_DTYPES[code]).newbyteorder(end)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    out = data.astype(np.float64)
    if slope not in (0.0, 1.0) or inter != 0.0:
        out = out * (slope if slope != 0.0 else 1.0) + inter
    aff = np.eye(4)
    if struct.unpack_from(end + "h", raw, 254)[0] > 0:
        aff[0] = struct.unpack_from(end + "4f", raw, 280)
        aff[1] = struct.unpack_from(end + "4f", raw, 296)
        aff[2] = struct.unpack_from(end + "4f", raw, 312)
    return out, aff
