"""Minimal TIFF reader / writer for the reference's on-disk data.

The reference moves every image through `iio.read` / `iio.write` (library.py:71-77) and skimage's
`imsave` (dataset/generate_raw_from_RGB.py:182): float32 4-channel packed raw frames, float32
2-channel flows, float32 3-channel results (`*_denoised.tif`), uint16 3-channel linear-RGB ground
truth.  Neither iio, tifffile nor libtiff bindings exist in this image and Pillow cannot open
multi-channel float TIFFs, so this is a reader for baseline TIFF 6.0 (+ BigTIFF) as those writers
produce it: strips or tiles, chunky or planar, little/big endian, uint/int/float samples of 8-64
bits, compression none / LZW / Deflate / PackBits, predictor none / horizontal / floating point.

`read(path)` returns what `iio.read` returns: an array [H, W, C] (C = 1 included) in the file's
sample type.  `write(path, arr)` writes uncompressed little-endian chunky TIFF in arr's dtype, which
libtiff-based readers (iio, tifffile) open.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
          9: ("i", 4), 10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


class TiffError(ValueError):
    pass


def _lzw_decode(data: bytes) -> bytes:
    """TIFF 6.0 section 13: MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, early change."""
    out = bytearray()
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    nbits, bitbuf, bitcnt = 9, 0, 0
    prev = None
    n, pos = len(data), 0
    while True:
        while bitcnt < nbits:
            if pos >= n:
                return bytes(out)
            bitbuf = (bitbuf << 8) | data[pos]
            pos += 1
            bitcnt += 8
        code = (bitbuf >> (bitcnt - nbits)) & ((1 << nbits) - 1)
        bitcnt -= nbits
        if code == 257:
            break
        if code == 256:
            del table[258:]
            nbits, prev = 9, None
            continue
        if prev is None:
            if code >= 256:
                raise TiffError("corrupt LZW stream")
            entry = table[code]
        else:
            if code < len(table):
                entry = table[code]
                table.append(prev + entry[:1])
            elif code == len(table):
                entry = prev + prev[:1]
                table.append(entry)
            else:
                raise TiffError("corrupt LZW stream")
            ln = len(table)
            if ln >= 4095:
                nbits = 12
            elif ln >= 2047:
                nbits = 12
            elif ln >= 1023:
                nbits = 11
            elif ln >= 511:
                nbits = 10
        out += entry
        prev = entry
    return bytes(out)


def _packbits_decode(data: bytes) -> bytes:
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        c = data[i]
        i += 1
        if c < 128:
            out += data[i:i + c + 1]
            i += c + 1
        elif c > 128:
            out += data[i:i + 1] * (257 - c)
            i += 1
    return bytes(out)


_native_lzw = None          # False once the library turned out to be unavailable


def _lzw(buf: bytes, expected: int) -> bytes:
    """LZW through the native decoder of librvdd_hip.so (host code, ~100x the pure-Python one) when the library
    loads; both decoders are exact, the Python one keeps TIFF reading independent of the build."""
    global _native_lzw
    if _native_lzw is None:
        try:
            from . import _lib
            _native_lzw = _lib.load().rvdd_tiff_lzw_decode
        except Exception:
            _native_lzw = False
    if _native_lzw and expected > 0:
        import ctypes
        out = ctypes.create_string_buffer(expected)
        n = _native_lzw(buf, len(buf), out, expected)
        if n >= 0:
            return out.raw[:n]
        # the stream produced more than the chunk holds (padding codes) or is corrupt: let the reference decoder say
    return _lzw_decode(buf)


def _decompress(buf: bytes, compression: int, expected: int = 0) -> bytes:
    if compression == 1:
        return buf
    if compression == 5:
        return _lzw(buf, expected)
    if compression in (8, 32946):
        return zlib.decompress(buf)
    if compression == 32773:
        return _packbits_decode(buf)
    raise TiffError(f"unsupported TIFF compression {compression}")


def _undo_predictor(block: np.ndarray, predictor: int, dtype: np.dtype, rows: int, cols: int, spp: int, bo: str) -> np.ndarray:
    """block: decoded bytes of `rows` x `cols` pixels with `spp` interleaved samples."""
    if predictor == 1:
        return np.frombuffer(block, dtype=dtype.newbyteorder(bo), count=rows * cols * spp).reshape(rows, cols, spp)
    if predictor == 2:      # horizontal differencing, per sample, modular arithmetic in the sample width
        a = np.frombuffer(block, dtype=dtype.newbyteorder(bo), count=rows * cols * spp).reshape(rows, cols, spp)
        u = a.astype(np.dtype(f"u{dtype.itemsize}"), copy=True) if dtype.kind in "iu" else None
        if u is None:
            raise TiffError("horizontal predictor on non-integer samples")
        return np.cumsum(u, axis=1, dtype=u.dtype).view(np.dtype(dtype.str[1:])).reshape(rows, cols, spp)
    if predictor == 3:      # floating point predictor (TIFF TechNote 3): byte planes, MSB first, byte-differenced
        bps = dtype.itemsize
        raw = np.frombuffer(block, dtype=np.uint8, count=rows * cols * spp * bps).reshape(rows, cols * spp * bps)
        # differencing runs over bytes with a stride of spp
        r = raw.reshape(rows, cols * bps, spp)
        r = np.cumsum(r, axis=1, dtype=np.uint8).reshape(rows, bps, cols * spp)
        be = np.ascontiguousarray(r.transpose(0, 2, 1))                      # [rows, cols*spp, bps] big-endian bytes
        return be.view(dtype.newbyteorder(">")).reshape(rows, cols, spp)
    raise TiffError(f"unsupported TIFF predictor {predictor}")


def read(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 8:
        raise TiffError(f"{path}: not a TIFF file")
    bo = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if bo is None:
        raise TiffError(f"{path}: not a TIFF file")
    magic = struct.unpack(bo + "H", buf[2:4])[0]
    if magic == 42:
        big, (ifd,) = False, struct.unpack(bo + "I", buf[4:8])
    elif magic == 43:
        big, (ifd,) = True, struct.unpack(bo + "Q", buf[8:16])
    else:
        raise TiffError(f"{path}: bad TIFF magic {magic}")
    cnt_fmt, cnt_sz, ent_sz, val_sz = ("Q", 8, 20, 8) if big else ("H", 2, 12, 4)
    (nent,) = struct.unpack(bo + cnt_fmt, buf[ifd:ifd + cnt_sz])
    tags = {}
    for i in range(nent):
        e = ifd + cnt_sz + i * ent_sz
        tag, typ = struct.unpack(bo + "HH", buf[e:e + 4])
        (count,) = struct.unpack(bo + ("Q" if big else "I"), buf[e + 4:e + 4 + val_sz])
        if typ not in _TYPES:
            continue
        fmt, size = _TYPES[typ]
        nbytes = size * count
        off = e + 4 + val_sz
        if nbytes > val_sz:
            (off,) = struct.unpack(bo + ("Q" if big else "I"), buf[off:off + val_sz])
        vals = struct.unpack(bo + fmt * count if len(fmt) == 1 else bo + fmt * count, buf[off:off + nbytes])
        tags[tag] = vals
    try:
        W, H = int(tags[256][0]), int(tags[257][0])
    except KeyError:
        raise TiffError(f"{path}: missing ImageWidth/ImageLength")
    spp = int(tags.get(277, (1,))[0])
    bits = tags.get(258, (1,) * spp)
    if len(set(bits)) != 1:
        raise TiffError(f"{path}: mixed BitsPerSample {bits}")
    bps = int(bits[0])
    fmt = int(tags.get(339, (1,))[0])
    kind = {1: "u", 2: "i", 3: "f", 4: "u"}.get(fmt)
    if kind is None or bps not in (8, 16, 32, 64) or (kind == "f" and bps < 32 and bps != 16):
        raise TiffError(f"{path}: unsupported sample type (SampleFormat {fmt}, {bps} bits)")
    dtype = np.dtype(f"{kind}{bps // 8}")
    compression = int(tags.get(259, (1,))[0])
    predictor = int(tags.get(317, (1,))[0])
    if compression not in (5, 8, 32946):
        predictor = 1                                  # libtiff: the Predictor tag only acts inside the LZW / Deflate codecs
    planar = int(tags.get(284, (1,))[0])
    planes = spp if planar == 2 else 1
    chunk_spp = 1 if planar == 2 else spp
    out = np.zeros((planes, H, W, chunk_spp), dtype=dtype)
    if 322 in tags:                                   # tiles
        tw, th = int(tags[322][0]), int(tags[323][0])
        offs, cnts = tags[324], tags[325]
        tx, ty = (W + tw - 1) // tw, (H + th - 1) // th
        for p in range(planes):
            for j in range(ty):
                for i in range(tx):
                    k = (p * ty + j) * tx + i
                    raw = _decompress(buf[offs[k]:offs[k] + cnts[k]], compression, th * tw * chunk_spp * dtype.itemsize)
                    blk = _undo_predictor(raw, predictor, dtype, th, tw, chunk_spp, bo)
                    y0, x0 = j * th, i * tw
                    out[p, y0:y0 + th, x0:x0 + tw] = blk[:H - y0, :W - x0]
    else:
        rps = min(int(tags.get(278, (H,))[0]), H)
        offs = tags.get(273)
        if offs is None:
            raise TiffError(f"{path}: missing StripOffsets")
        nstrips = (H + rps - 1) // rps
        cnts = tags.get(279)
        if cnts is None:                               # only legal uncompressed
            cnts = tuple(min(rps, H - s * rps) * W * chunk_spp * dtype.itemsize for s in range(nstrips)) * planes
        for p in range(planes):
            for s in range(nstrips):
                k = p * nstrips + s
                rows = min(rps, H - s * rps)
                raw = _decompress(buf[offs[k]:offs[k] + cnts[k]], compression, rows * W * chunk_spp * dtype.itemsize)
                if len(raw) < rows * W * chunk_spp * dtype.itemsize:
                    raise TiffError(f"{path}: strip {k} is truncated")
                out[p, s * rps:s * rps + rows] = _undo_predictor(raw, predictor, dtype, rows, W, chunk_spp, bo)
    if planar == 2:
        return np.ascontiguousarray(out[..., 0].transpose(1, 2, 0))
    return out[0]


def write(path: str, arr) -> None:
    """[H,W] or [H,W,C] array of u8/u16/u32/i8/i16/i32/f32/f64 -> uncompressed little-endian TIFF."""
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3:
        raise TiffError("write: array must be [H,W] or [H,W,C]")
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype.kind not in "uif" or a.dtype.itemsize not in (1, 2, 4, 8) or (a.dtype.kind == "f" and a.dtype.itemsize < 4):
        raise TiffError(f"write: unsupported dtype {a.dtype}")
    H, W, C = a.shape
    data = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False)).tobytes()
    bps = a.dtype.itemsize * 8
    fmt = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
    entries = []                                        # (tag, type, count, values)

    def add(tag, typ, vals):
        entries.append((tag, typ, len(vals), vals))

    add(256, 4, (W,))
    add(257, 4, (H,))
    add(258, 3, (bps,) * C)
    add(259, 3, (1,))
    add(262, 3, (2 if C == 3 else 1,))
    add(273, 4, (0,))                                   # patched below
    add(277, 3, (C,))
    add(278, 4, (H,))
    add(279, 4, (len(data),))
    add(284, 3, (1,))
    extra = C - 3 if C > 3 else (C - 1 if C != 3 else 0)
    if extra > 0:
        add(338, 3, (0,) * extra)                       # EXTRASAMPLE_UNSPECIFIED
    add(339, 3, (fmt,) * C)
    if 8 + len(data) + 2 + 12 * len(entries) + 4 + 64 >= 2 ** 32:
        raise TiffError("write: image too large for classic TIFF")
    data_off = 8
    ifd_off = data_off + len(data) + (len(data) & 1)
    extra_off = ifd_off + 2 + 12 * len(entries) + 4
    ifd = struct.pack("<H", len(entries))
    tail = b""
    for tag, typ, count, vals in entries:
        if tag == 273:
            vals = (data_off,)
        f, size = _TYPES[typ]
        packed = struct.pack("<" + f * count, *vals)
        if len(packed) <= 4:
            field = packed.ljust(4, b"\0")
        else:
            field = struct.pack("<I", extra_off + len(tail))
            tail += packed + (b"\0" if len(packed) & 1 else b"")
        ifd += struct.pack("<HHI", tag, typ, count) + field
    ifd += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        f.write(data)
        if len(data) & 1:
            f.write(b"\0")
        f.write(ifd)
        f.write(tail)
