"""Blosc-1 container around zstd blocks: the chunk format the reference's store uses,
`zarr.Blosc(cname='zstd', clevel=3, shuffle=1)` (chunking_data.py:180).

numcodecs / c-blosc are not installed here; the zstd codec that ships inside pyarrow is.  This module writes (and
reads back) the published Blosc-1 frame by hand, so that a store written here opens in a stock zarr + numcodecs:

    header (16 B)  version 2 | versionlz 1 | flags | typesize | nbytes u32 | blocksize u32 | cbytes u32   (little endian)
                   flags: 0x01 byte shuffle, 0x02 stored uncompressed ("memcpyed"), 0x10 blocks are not split into
                   per-byte streams (what c-blosc >= 1.14 sets for zstd), 0x80 = compressor format 4 (zstd) << 5
    bstarts        one i32 per block: offset of the block from the start of the frame
    block          i32 csize + csize bytes: one zstd frame of the byte-shuffled block, or the shuffled bytes themselves
                   when csize == the block's length (c-blosc stores a block that zstd cannot shrink that way)

Block size and zstd level follow c-blosc's own rules for this configuration (compute_blocksize: 32 KiB x 2 for a
high-compression-ratio codec x 2 for clevel 3 = 128 KiB, a multiple of the type size; zstd level 2·clevel − 1), a buffer
under 128 bytes or one that does not shrink is stored with the 0x02 flag.  The compressed bits may differ from
c-blosc's (another zstd release), the frame does not.  Parity: unpinned — no Blosc decoder exists in this image to read
a frame back, so the tests check the header fields against the format and round-trip through the reader below.
"""
import struct

import numpy as np

_VERSION_FORMAT = 2
_ZSTD_VERSION_FORMAT = 1
_F_SHUFFLE, _F_MEMCPY, _F_BITSHUFFLE, _F_DONT_SPLIT = 0x01, 0x02, 0x04, 0x10
_ZSTD_FORMAT = 4
_MAX_OVERHEAD = 16
_MIN_BUFFERSIZE = 128
_L1 = 32 * 1024
_MAX_SPLITS = 16

_codecs = {}


def _zstd(level):
    c = _codecs.get(level)
    if c is None:
        import pyarrow as pa
        if not pa.Codec.is_available("zstd"):
            raise RuntimeError("compressor 'blosc' needs a zstd codec (pyarrow's is not available)")
        c = _codecs[level] = pa.Codec("zstd", compression_level=level)
    return c


def available():
    try:
        _zstd(5)
        return True
    except Exception:
        return False


def blocksize_for(nbytes, typesize, clevel):
    """c-blosc compute_blocksize() for a high-compression-ratio codec that is not split (zstd), automatic block size."""
    if nbytes < typesize:
        return 1
    bs = nbytes
    if nbytes >= _L1:
        bs = _L1 * 2
        bs = {0: bs // 4, 1: bs // 2, 2: bs, 3: bs * 2, 4: bs * 4, 5: bs * 4, 6: bs * 8, 7: bs * 8, 8: bs * 8,
              9: bs * 16}[int(clevel)]
    bs = min(bs, nbytes)
    if bs > typesize:
        bs = bs // typesize * typesize
    return bs


def _shuffle(buf, typesize):
    n = len(buf) // typesize
    a = np.frombuffer(buf, dtype=np.uint8)
    out = np.empty(len(buf), dtype=np.uint8)
    out[:n * typesize] = a[:n * typesize].reshape(n, typesize).T.reshape(-1)
    out[n * typesize:] = a[n * typesize:]
    return out.tobytes()


def _unshuffle(buf, typesize):
    n = len(buf) // typesize
    a = np.frombuffer(buf, dtype=np.uint8)
    out = np.empty(len(buf), dtype=np.uint8)
    out[:n * typesize] = a[:n * typesize].reshape(typesize, n).T.reshape(-1)
    out[n * typesize:] = a[n * typesize:]
    return out.tobytes()


def compress(raw, typesize, clevel=3, shuffle=1):
    """One Blosc-1 frame (cname zstd) of `raw` (bytes-like)."""
    raw = bytes(raw)
    nbytes = len(raw)
    if not 1 <= typesize <= 255:
        typesize = 1
    if nbytes > 0x7fffffff - _MAX_OVERHEAD:
        raise ValueError("a Blosc-1 frame holds less than 2 GiB")
    flags = (_ZSTD_FORMAT << 5) | _F_DONT_SPLIT | (_F_SHUFFLE if shuffle == 1 else 0)
    if shuffle not in (0, 1):
        raise ValueError("shuffle must be 0 (none) or 1 (byte shuffle)")
    blocksize = blocksize_for(nbytes, typesize, clevel)

    def memcpyed():
        return struct.pack("<BBBBIII", _VERSION_FORMAT, _ZSTD_VERSION_FORMAT, flags | _F_MEMCPY, typesize, nbytes,
                           blocksize, nbytes + _MAX_OVERHEAD) + raw

    if nbytes < _MIN_BUFFERSIZE or clevel == 0:
        return memcpyed()
    codec = _zstd(2 * int(clevel) - 1 if clevel < 9 else 22)
    nblocks = -(-nbytes // blocksize)
    pos = _MAX_OVERHEAD + 4 * nblocks
    bstarts, parts = [], []
    for b in range(nblocks):
        blk = raw[b * blocksize:(b + 1) * blocksize]
        if shuffle == 1 and typesize > 1:
            blk = _shuffle(blk, typesize)
        c = codec.compress(blk, asbytes=True)              # releases the GIL
        if len(c) >= len(blk):
            c = blk                                        # csize == block length: stored as is
        bstarts.append(pos)
        parts.append(struct.pack("<i", len(c)))
        parts.append(c)
        pos += 4 + len(c)
        if pos > nbytes + _MAX_OVERHEAD:                   # c-blosc gives up and copies when the frame would outgrow its bound
            return memcpyed()
    head = struct.pack("<BBBBIII", _VERSION_FORMAT, _ZSTD_VERSION_FORMAT, flags, typesize, nbytes, blocksize, pos)
    return b"".join([head, struct.pack("<%di" % nblocks, *bstarts)] + parts)


def decompress(frame):
    """The bytes of one Blosc-1 frame whose blocks are zstd frames (or stored)."""
    frame = bytes(frame)
    if len(frame) < _MAX_OVERHEAD:
        raise ValueError("not a Blosc frame: shorter than its header")
    version, _vlz, flags, typesize, nbytes, blocksize, cbytes = struct.unpack_from("<BBBBIII", frame, 0)
    if version != _VERSION_FORMAT:
        raise ValueError(f"Blosc frame format {version} is not supported")
    if cbytes != len(frame):
        raise ValueError(f"Blosc frame says {cbytes} bytes, the chunk has {len(frame)}")
    if flags & _F_MEMCPY:
        return frame[_MAX_OVERHEAD:_MAX_OVERHEAD + nbytes]
    if (flags >> 5) != _ZSTD_FORMAT:
        raise ValueError(f"Blosc compressor format {flags >> 5} is not zstd; only zstd frames can be read here")
    if flags & _F_BITSHUFFLE:
        raise ValueError("bit-shuffled Blosc frames are not supported")
    if nbytes == 0:
        return b""
    codec = _zstd(5)
    nblocks = -(-nbytes // blocksize)
    bstarts = struct.unpack_from("<%di" % nblocks, frame, _MAX_OVERHEAD)
    out = []
    for b in range(nblocks):
        neblock = min(blocksize, nbytes - b * blocksize)
        leftover = neblock != blocksize
        split = (not (flags & _F_DONT_SPLIT) and typesize <= _MAX_SPLITS and blocksize // typesize >= _MIN_BUFFERSIZE
                 and not leftover)
        nstreams = typesize if split else 1
        ne = neblock // nstreams
        pos = bstarts[b]
        streams = []
        for _ in range(nstreams):
            (csize,) = struct.unpack_from("<i", frame, pos)
            pos += 4
            if csize < 0 or pos + csize > len(frame):
                raise ValueError("corrupt Blosc frame: a block runs past the end")
            data = frame[pos:pos + csize]
            pos += csize
            streams.append(data if csize == ne else codec.decompress(data, decompressed_size=ne, asbytes=True))
        blk = b"".join(streams)
        if len(blk) != neblock:
            raise ValueError("corrupt Blosc frame: a block has the wrong length")
        if (flags & _F_SHUFFLE) and typesize > 1:
            blk = _unshuffle(blk, typesize)
        out.append(blk)
    return b"".join(out)
