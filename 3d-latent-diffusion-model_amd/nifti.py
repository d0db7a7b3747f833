"""Minimal NIfTI-1 (.nii) writer: what `nib.save(nib.Nifti1Image(vol, np.eye(4)), filename)` produces for a float32
volume at 3d_ldm/inference.py:100-102 (nibabel is not available here).  348-byte header + 4 bytes extension flag + data
in Fortran (x fastest) order, identity sform."""
from __future__ import annotations

import struct

import numpy as np


def save_nifti(volume: np.ndarray, filename: str) -> str:
    vol = np.asarray(volume, dtype=np.float32)
    if vol.ndim < 3 or vol.ndim > 5:
        raise ValueError("expected a 3-5 dimensional volume")
    if not (filename.endswith(".nii") or filename.endswith(".nii.gz")):
        filename += ".nii"
    dims = [vol.ndim] + list(vol.shape) + [1] * (7 - vol.ndim)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)                      # sizeof_hdr
    struct.pack_into("<8h", hdr, 40, *dims)                  # dim[8]
    struct.pack_into("<h", hdr, 70, 16)                      # datatype FLOAT32
    struct.pack_into("<h", hdr, 72, 32)                      # bitpix
    struct.pack_into("<8f", hdr, 76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)   # pixdim
    struct.pack_into("<f", hdr, 108, 352.0)                  # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)                    # scl_slope
    struct.pack_into("<h", hdr, 254, 2)                      # sform_code = aligned
    struct.pack_into("<4f", hdr, 280, 1.0, 0.0, 0.0, 0.0)    # srow_x
    struct.pack_into("<4f", hdr, 296, 0.0, 1.0, 0.0, 0.0)    # srow_y
    struct.pack_into("<4f", hdr, 312, 0.0, 0.0, 1.0, 0.0)    # srow_z
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + b"\0\0\0\0" + np.asfortranarray(vol).tobytes(order="F")
    if filename.endswith(".gz"):
        import gzip
        with gzip.open(filename, "wb") as f:
            f.write(payload)
    else:
        with open(filename, "wb") as f:
            f.write(payload)
    return filename
