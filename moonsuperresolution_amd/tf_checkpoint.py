"""Weight import from Keras SavedModel / TF checkpoints without TensorFlow.

The reference stores its networks as Keras SavedModel directories — ``gaugan.save(dir)`` writes
``generator/ discriminator/ encoder/`` (spade/models/model.py:569-605) and inference reads them back with
``gaugan.load(path+'generator', ...)`` (process_full_tiles.py:30, model.py:607-610).  The tensors of a SavedModel
live in ``<dir>/variables/variables.index`` + ``variables.data-00000-of-00001``: a TensorBundle.  This module reads
(and, for tests, writes) that format from its published layout:

* ``.index`` is a LevelDB-style sorted string table: data blocks of prefix-compressed (key, value) entries with a
  restart array, each block followed by a 5-byte trailer (compression type + masked CRC32C), then a metaindex block,
  an index block and a 48-byte footer (two block handles, padding, magic 0xdb4775248b80fb57).
* key "" holds a BundleHeaderProto; every other key is a tensor name whose value is a BundleEntryProto
  {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
* ``.data-XXXXX-of-NNNNN`` shards hold the raw little-endian tensor bytes at (offset, size).

NOT validated against a TensorFlow-written file: none ships with the reference (no weights are published) and
TensorFlow cannot be installed here; the round trip reader <-> writer of this module and the CRCs are what
tests/test_tf_checkpoint.py checks.  Snappy-compressed index blocks (TF writes them uncompressed) are rejected.

``keras_to_weights`` maps Keras object-graph keys such as
``layer_with_weights-4/spade_3/conv_gamma/kernel/.ATTRIBUTES/VARIABLE_VALUE`` to this package's names
(``gen.rb4.spade_3.conv_gamma.kernel``), following the layer order of build_generator / build_encoder
(spade/models/networks.py:8-57) and the attribute names of ResidualBlock / SPADE (blocks.py:16-26, spade.py:9-11).
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Iterable, List, Mapping, Optional, Tuple

import numpy as np

from .weights import weight_shapes

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_STRING, DT_INT64, DT_INT32 = 1, 7, 9, 3
_NP_OF_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 6: np.int8, 10: np.bool_}
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


# ---- CRC32C (Castagnoli), masked the LevelDB way ----------------------------------------------------------------
def _make_crc_table():
    poly = 0x82F63B78
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_CRC_TABLE = _make_crc_table()


def _native_crc():
    """msr_crc32c of libmoonsr_hip.so (host code; loads without a GPU) or None."""
    try:
        from . import _lib
        return _lib.load().msr_crc32c
    except Exception:
        return None


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of a bytes-like object; uses the native helper for anything large."""
    if len(data) >= 4096:
        fn = _native_crc()
        if fn is not None:
            buf = np.frombuffer(data, dtype=np.uint8)
            return int(fn(buf.ctypes.data, buf.size, crc))
    data = bytes(data)
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints / tiny protobuf helpers ---------------------------------------------------------------------------------
def _put_varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    v = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _pb_fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    """Yield (field number, wire type, value) of a serialized protobuf message (varint / 64-bit / bytes / 32-bit)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _pb_key(fn: int, wt: int) -> bytes:
    return _put_varint((fn << 3) | wt)


def _entry_proto(dtype: int, shape: Tuple[int, ...], shard: int, offset: int, size: int, crc: int) -> bytes:
    dims = b"".join(_pb_key(2, 2) + _put_varint(len(d)) + d for d in
                    (_pb_key(1, 0) + _put_varint(s) for s in shape))
    msg = _pb_key(1, 0) + _put_varint(dtype)
    msg += _pb_key(2, 2) + _put_varint(len(dims)) + dims
    if shard:
        msg += _pb_key(3, 0) + _put_varint(shard)
    if offset:
        msg += _pb_key(4, 0) + _put_varint(offset)
    msg += _pb_key(5, 0) + _put_varint(size)
    msg += _pb_key(6, 5) + struct.pack("<I", crc)
    return msg


def _parse_entry(buf: bytes):
    dtype = shard = offset = size = crc = 0
    shape: List[int] = []
    for fn, _, v in _pb_fields(buf):
        if fn == 1:
            dtype = v
        elif fn == 2:
            for f2, _, d in _pb_fields(v):
                if f2 == 2:
                    size_d = 0
                    for f3, _, dv in _pb_fields(d):
                        if f3 == 1:
                            size_d = dv
                    shape.append(size_d)
        elif fn == 3:
            shard = v
        elif fn == 4:
            offset = v
        elif fn == 5:
            size = v
        elif fn == 6:
            crc = v
    return dtype, tuple(shape), shard, offset, size, crc


# ---- table blocks --------------------------------------------------------------------------------------------------------
def _parse_block(block: bytes) -> List[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    out, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def _read_block(f: bytes, offset: int, size: int, verify: bool) -> bytes:
    block = f[offset:offset + size]
    ctype = f[offset + size]
    if verify:
        stored = struct.unpack_from("<I", f, offset + size + 1)[0]
        if mask_crc(crc32c(f[offset:offset + size + 1])) != stored:
            raise ValueError("index block CRC mismatch")
    if ctype != 0:
        raise ValueError("compressed index blocks are not supported (TensorFlow writes them uncompressed)")
    return block


def _build_block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts, last = [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _handle(offset: int, size: int) -> bytes:
    return _put_varint(offset) + _put_varint(size)


# ---- public: read / write a bundle -------------------------------------------------------------------------------------
def list_tensor_bundle(prefix: str, verify_crc: bool = True) -> Dict[str, Tuple[int, Tuple[int, ...], int, int, int, int]]:
    """name -> (dtype, shape, shard_id, offset, size, crc32c) of every tensor in ``prefix.index``."""
    with open(prefix + ".index", "rb") as fh:
        f = fh.read()
    if len(f) < 48 or struct.unpack_from("<Q", f, len(f) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{prefix}.index is not a TensorBundle index (bad magic)")
    footer = f[-48:]
    _, p = _get_varint(footer, 0)          # metaindex offset
    _, p = _get_varint(footer, p)          # metaindex size
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    entries = {}
    for _, hv in _parse_block(_read_block(f, ioff, isize, verify_crc)):
        boff, q = _get_varint(hv, 0)
        bsize, q = _get_varint(hv, q)
        for k, v in _parse_block(_read_block(f, boff, bsize, verify_crc)):
            if k == b"":
                continue                    # BundleHeaderProto
            entries[k.decode()] = _parse_entry(v)
    return entries


def read_tensor_bundle(prefix: str, names: Optional[Iterable[str]] = None, verify_crc: bool = False) -> Dict[str, np.ndarray]:
    """Read tensors (all numeric ones, or ``names``) of a TensorBundle ``prefix`` (.index + .data-*) as NumPy arrays."""
    entries = list_tensor_bundle(prefix)
    n_shards = 1 + max((e[2] for e in entries.values()), default=0)
    want = set(names) if names is not None else None
    out: Dict[str, np.ndarray] = {}
    shards: Dict[int, np.memmap] = {}
    for name, (dtype, shape, shard, offset, size, crc) in entries.items():
        if want is not None and name not in want:
            continue
        if dtype not in _NP_OF_DT:
            if want is not None:
                raise ValueError(f"{name}: unsupported dtype enum {dtype}")
            continue                        # e.g. the _CHECKPOINTABLE_OBJECT_GRAPH string tensor
        if shard not in shards:
            shards[shard] = np.memmap(f"{prefix}.data-{shard:05d}-of-{n_shards:05d}", dtype=np.uint8, mode="r")
        raw = shards[shard][offset:offset + size]
        if verify_crc and mask_crc(crc32c(np.ascontiguousarray(raw))) != crc:
            raise ValueError(f"{name}: data CRC mismatch")
        arr = np.frombuffer(raw.tobytes(), dtype=np.dtype(_NP_OF_DT[dtype]).newbyteorder("<")).reshape(shape)
        out[name] = arr
    if want is not None:
        missing = want - set(out)
        if missing:
            raise KeyError(f"tensors not in the bundle: {sorted(missing)[:5]}")
    return out


def write_tensor_bundle(prefix: str, tensors: Mapping[str, np.ndarray], block_entries: int = 64) -> None:
    """Write ``tensors`` as a single-shard TensorBundle (used by the tests; float32 / int arrays only)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: List[Tuple[bytes, bytes]] = []
    header = _pb_key(1, 0) + _put_varint(1)            # BundleHeaderProto.num_shards = 1 (little endian = default)
    header += _pb_key(3, 2) + _put_varint(2) + _pb_key(1, 0) + _put_varint(1)    # version { producer: 1 }
    items.append((b"", header))
    offset = 0
    with open(f"{prefix}.data-00000-of-00001", "wb") as data:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])
            if a.ndim and not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            dt = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int32): DT_INT32, np.dtype(np.int64): DT_INT64}[a.dtype]
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            data.write(raw)
            items.append((name.encode(), _entry_proto(dt, a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    items.sort(key=lambda kv: kv[0])
    out = bytearray()
    index_entries: List[Tuple[bytes, bytes]] = []

    def emit(block: bytes) -> Tuple[int, int]:
        off = len(out)
        out.extend(block)
        out.append(0)                                   # no compression
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        off, size = emit(_build_block(chunk))
        index_entries.append((chunk[-1][0], _handle(off, size)))
    moff, msize = emit(_build_block([]))
    ioff, isize = emit(_build_block(index_entries, restart_interval=1))
    footer = _handle(moff, msize) + _handle(ioff, isize)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as fh:
        fh.write(bytes(out))


# ---- Keras object-graph keys -> package weight names -----------------------------------------------------------------
def _strip(key: str) -> Optional[str]:
    if not key.endswith(_SUFFIX) or "/.OPTIMIZER_SLOT/" in key or key.startswith("optimizer"):
        return None
    return key[:-len(_SUFFIX)]


def generator_key_to_name(key: str) -> Optional[str]:
    """build_generator (networks.py:37-57): layer_with_weights-0 = Dense, -1..-6 = ResidualBlocks, -7 = head Conv2D."""
    k = _strip(key)
    if k is None:
        return None
    m = re.fullmatch(r"layer_with_weights-(\d+)/(.+)", k)
    if not m:
        return None
    i, rest = int(m.group(1)), m.group(2)
    if i == 0 and rest in ("kernel", "bias"):
        return f"gen.dense.{rest}"
    if i == 7 and rest in ("kernel", "bias"):
        return f"gen.head.{rest}"
    if 1 <= i <= 6:
        m2 = re.fullmatch(r"(spade_[123])/(conv|conv_gamma|conv_beta)/(kernel|bias)", rest)
        if m2:
            return f"gen.rb{i}.{m2.group(1)}.{m2.group(2)}.{m2.group(3)}"
        m2 = re.fullmatch(r"(conv_[123])/(kernel|bias)", rest)
        if m2:
            return f"gen.rb{i}.{m2.group(1)}.{m2.group(2)}"
    return None


def encoder_key_to_name(key: str) -> Optional[str]:
    """build_encoder (networks.py:8-34): layer_with_weights-0..4 = downsample Sequentials (conv = their weighted
    layer 0, InstanceNormalization = weighted layer 1), -5 = Dense 'mean', -6 = Dense 'variance'."""
    k = _strip(key)
    if k is None:
        return None
    m = re.fullmatch(r"layer_with_weights-(\d+)/(.+)", k)
    if not m:
        return None
    i, rest = int(m.group(1)), m.group(2)
    if i == 5 and rest in ("kernel", "bias"):
        return f"enc.mean.{rest}"
    if i == 6 and rest in ("kernel", "bias"):
        return f"enc.variance.{rest}"
    if 0 <= i <= 4:
        if rest == "layer_with_weights-0/kernel":
            return f"enc.ds{i + 1}.kernel"
        m2 = re.fullmatch(r"layer_with_weights-1/(gamma|beta)", rest)
        if m2 and i >= 1:
            return f"enc.ds{i + 1}.in.{m2.group(1)}"
    return None


def keras_to_weights(generator_dir: str, encoder_dir: str, image_size: int, latent_dim: int = 256,
                     name_map: Optional[Mapping[str, str]] = None) -> Dict[str, np.ndarray]:
    """Read the ``generator/`` and ``encoder/`` SavedModel directories written by ``GauGAN.save``
    (spade/models/model.py:569-605) into the name -> array dict ``Generator.load`` takes.

    ``name_map`` (checkpoint key -> package name) overrides the built-in rules for models saved with a different
    object-graph layout.  Raises ValueError listing what is missing / mis-shaped."""
    expected = weight_shapes("gaugan", image_size, latent_dim)
    out: Dict[str, np.ndarray] = {}
    unmatched: List[str] = []
    for d, rule in ((generator_dir, generator_key_to_name), (encoder_dir, encoder_key_to_name)):
        prefix = os.path.join(d, "variables", "variables")
        if not os.path.exists(prefix + ".index"):
            raise ValueError(f"{prefix}.index not found: '{d}' is not a Keras SavedModel directory")
        for key, arr in read_tensor_bundle(prefix).items():
            name = (name_map or {}).get(key) or rule(key)
            if name is None:
                unmatched.append(key)
                continue
            out[name] = np.ascontiguousarray(arr, dtype=np.float32)
    missing = [n for n in expected if n not in out]
    bad = [n for n in expected if n in out and tuple(out[n].shape) != tuple(expected[n])]
    if missing or bad:
        raise ValueError(f"SavedModel does not match GauGAN({image_size}): missing {missing[:6]}, wrong shape {bad[:6]}; "
                         f"unmatched checkpoint keys {unmatched[:6]} (pass name_map to override the key rules)")
    return {n: out[n] for n in expected}
