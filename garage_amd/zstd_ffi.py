"""zstd through the system's libzstd.so.1 (no headers / Python module in this
image, but the runtime library is there): what DataBlock::from_buffer and
zstd_encode do in the reference -- one-shot frame, compression level from the
config, **content checksum enabled** (src/block/block.rs:99-106)."""
from __future__ import annotations

import ctypes
import ctypes.util

ZSTD_c_compressionLevel = 100
ZSTD_c_checksumFlag = 201
_CONTENTSIZE_UNKNOWN = (1 << 64) - 1
_CONTENTSIZE_ERROR = (1 << 64) - 2


def _load():
    for name in ("libzstd.so.1", ctypes.util.find_library("zstd")):
        if not name:
            continue
        try:
            z = ctypes.CDLL(name)
        except OSError:
            continue
        sz, vp = ctypes.c_size_t, ctypes.c_void_p
        z.ZSTD_createCCtx.restype = vp
        z.ZSTD_freeCCtx.argtypes = [vp]
        z.ZSTD_CCtx_setParameter.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        z.ZSTD_CCtx_setParameter.restype = sz
        z.ZSTD_compress2.argtypes = [vp, vp, sz, vp, sz]
        z.ZSTD_compress2.restype = sz
        z.ZSTD_compressBound.argtypes = [sz]
        z.ZSTD_compressBound.restype = sz
        z.ZSTD_decompress.argtypes = [vp, sz, vp, sz]
        z.ZSTD_decompress.restype = sz
        z.ZSTD_getFrameContentSize.argtypes = [vp, sz]
        z.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        z.ZSTD_isError.argtypes = [sz]
        z.ZSTD_isError.restype = ctypes.c_uint
        z.ZSTD_getErrorName.argtypes = [sz]
        z.ZSTD_getErrorName.restype = ctypes.c_char_p
        return z
    return None


_z = _load()


def available() -> bool:
    return _z is not None


def zstd_encode(data: bytes, level: int) -> bytes:
    if _z is None:
        raise RuntimeError("libzstd is not available")
    cctx = _z.ZSTD_createCCtx()
    if not cctx:
        raise MemoryError("ZSTD_createCCtx")
    try:
        for p, v in ((ZSTD_c_compressionLevel, level), (ZSTD_c_checksumFlag, 1)):
            rc = _z.ZSTD_CCtx_setParameter(cctx, p, v)
            if _z.ZSTD_isError(rc):
                raise ValueError(_z.ZSTD_getErrorName(rc).decode())
        cap = _z.ZSTD_compressBound(len(data))
        out = ctypes.create_string_buffer(cap)
        n = _z.ZSTD_compress2(cctx, out, cap, data, len(data))
        if _z.ZSTD_isError(n):
            raise ValueError(_z.ZSTD_getErrorName(n).decode())
        return out.raw[:n]
    finally:
        _z.ZSTD_freeCCtx(cctx)


def zstd_decode(data: bytes, max_len: int = 1 << 30) -> bytes:
    """Decodes one frame; raises ValueError on corruption (the frame checksum is
    verified), which is what DataBlock::verify relies on for compressed blocks."""
    if _z is None:
        raise RuntimeError("libzstd is not available")
    size = _z.ZSTD_getFrameContentSize(data, len(data))
    if size in (_CONTENTSIZE_ERROR, _CONTENTSIZE_UNKNOWN) or size > max_len:
        raise ValueError("not a zstd frame with a known content size")
    out = ctypes.create_string_buffer(max(size, 1))
    n = _z.ZSTD_decompress(out, size, data, len(data))
    if _z.ZSTD_isError(n):
        raise ValueError(_z.ZSTD_getErrorName(n).decode())
    if n != size:
        raise ValueError("zstd: decoded size mismatch")
    return out.raw[:n]
