"""IQ windows on the wire: the two formats either side of the engine.  Host-side only.

1. What the reference sends today (iq_stream_client.py:31-44, :237, :296-312): a detection's `iq_samples` is
   `ndarray.tolist()` -- Python complex numbers -- pushed through `NumpyEncoder`, which turns every complex into
   `str(obj)`, e.g. "(12.5-3.5j)"; the server keeps the list of strings as it arrived (central_processor.py:54,
   :405-414 drops it before tdoa_processor).  `parse_complex_list` turns such a list (strings, complex numbers or
   [re, im] pairs, mixed) into the complex64 window the engine takes.

2. A binary frame for whole capture windows (the reference has none: 256 JSON strings per detection are ~6 KB for
   2 KB of samples, and a 16 384-sample capture would be 400 KB of text).  One frame = fixed little-endian header +
   node id + payload, payload either the raw rtl_sdr bytes (uint8 I,Q interleaved, buoy_node.py:392-398: 2 B/sample,
   what `RMX_IN_U8` ingests without a host pass) or complex64.  `pack_iq_frame` / `unpack_iq_frame`; `frames_to_batch`
   stacks the frames of one (frequency, window) group in buoy order into the [B][N] array of `TDoACalculator`.

No sockets here: transport is the caller's (SURVEY.md section 8 marks networking out of scope).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Iterable, List, Sequence

import numpy as np

MAGIC = b"RMXQ"
VERSION = 1
DT_U8_IQ = 0        # uint8 I,Q interleaved: 2 bytes per sample (rtl_sdr's own format)
DT_COMPLEX64 = 1    # float32 I,Q interleaved: 8 bytes per sample
# magic, version, dtype, flags, node id length, n_samples, sample rate, centre frequency, gps timestamp
_HEADER = struct.Struct("<4sBBHHxxIddq")


class IqWireError(ValueError):
    pass


def parse_complex_list(items: Iterable) -> np.ndarray:
    """The reference's JSON encoding of an IQ excerpt -> complex64 [N].

    Accepts what `json.loads` hands back for a list the reference's `NumpyEncoder` wrote -- strings such as
    "(1.5-2j)", "3j", "(nan+0j)" -- as well as Python / numpy complex numbers and [re, im] pairs."""
    out: List[complex] = []
    for k, it in enumerate(items):
        if isinstance(it, str):
            try:
                out.append(complex(it.strip()))      # str(complex) round-trips through complex()
            except ValueError as e:
                raise IqWireError(f"iq_samples[{k}] = {it!r} is not a complex literal") from e
        elif isinstance(it, (list, tuple)) and len(it) == 2:
            out.append(complex(float(it[0]), float(it[1])))
        elif isinstance(it, (complex, float, int, np.number)):
            out.append(complex(it))
        else:
            raise IqWireError(f"iq_samples[{k}] has unsupported type {type(it).__name__}")
    return np.asarray(out, dtype=np.complex64)


@dataclass
class IqFrame:
    node_id: str
    sample_rate_hz: float
    center_freq_hz: float
    gps_timestamp_ns: int
    samples: np.ndarray          # uint8 [2N] (DT_U8_IQ) or complex64 [N]

    @property
    def n_samples(self) -> int:
        return self.samples.shape[0] // 2 if self.samples.dtype == np.uint8 else self.samples.shape[0]


def pack_iq_frame(node_id: str, sample_rate_hz: float, center_freq_hz: float, gps_timestamp_ns: int, samples) -> bytes:
    """One capture window as bytes.  `samples`: uint8 [2N] (raw rtl_sdr bytes) or anything complex -> complex64 [N]."""
    a = np.asarray(samples)
    if a.ndim != 1:
        raise IqWireError(f"samples must be one-dimensional, got shape {a.shape}")
    if a.dtype == np.uint8:
        if a.shape[0] % 2:
            raise IqWireError("uint8 IQ needs an even number of bytes (I,Q interleaved)")
        dt, n = DT_U8_IQ, a.shape[0] // 2
    else:
        a = np.ascontiguousarray(a, dtype=np.complex64)
        dt, n = DT_COMPLEX64, a.shape[0]
    nid = node_id.encode("utf-8")
    if len(nid) > 0xFFFF:
        raise IqWireError("node id longer than 65535 bytes")
    head = _HEADER.pack(MAGIC, VERSION, dt, 0, len(nid), n, float(sample_rate_hz), float(center_freq_hz), int(gps_timestamp_ns))
    return head + nid + np.ascontiguousarray(a).tobytes()


def unpack_iq_frame(buf) -> IqFrame:
    """Inverse of `pack_iq_frame`; every length is checked against the buffer before anything is sliced."""
    mv = memoryview(buf).cast("B")
    if len(mv) < _HEADER.size:
        raise IqWireError(f"frame of {len(mv)} bytes is shorter than the {_HEADER.size}-byte header")
    magic, ver, dt, _flags, nid_len, n, fs, fc, ts = _HEADER.unpack_from(mv, 0)
    if magic != MAGIC:
        raise IqWireError(f"bad magic {bytes(magic)!r}")
    if ver != VERSION:
        raise IqWireError(f"frame version {ver}, this reader knows {VERSION}")
    if dt not in (DT_U8_IQ, DT_COMPLEX64):
        raise IqWireError(f"unknown sample type {dt}")
    per = 2 if dt == DT_U8_IQ else 8
    need = _HEADER.size + nid_len + n * per
    if len(mv) != need:
        raise IqWireError(f"frame is {len(mv)} bytes, header says {need}")
    off = _HEADER.size
    try:
        node_id = bytes(mv[off:off + nid_len]).decode("utf-8")
    except UnicodeDecodeError as e:
        raise IqWireError("node id is not UTF-8") from e
    off += nid_len
    if dt == DT_U8_IQ:
        samples = np.frombuffer(mv, dtype=np.uint8, count=2 * n, offset=off).copy()
    else:
        samples = np.frombuffer(mv, dtype=np.complex64, count=n, offset=off).copy()
    return IqFrame(node_id, fs, fc, ts, samples)


def frames_to_batch(frames: Sequence[IqFrame], buoy_order: Sequence[str]) -> np.ndarray:
    """[B][N] complex64 (or [B][2N] uint8) of one group, rows in `buoy_order` (= the detection-list order that fixes
    the pair order, tdoa_processor.py:156-157).  All frames must agree in type, length and sample rate: the same rule
    as the calculator's seam (partial or mismatched IQ yields no measurements)."""
    by_id = {}
    for f in frames:
        if f.node_id in by_id:
            raise IqWireError(f"two frames from node {f.node_id!r} in one group")
        by_id[f.node_id] = f
    missing = [b for b in buoy_order if b not in by_id]
    if missing:
        raise IqWireError(f"no frame from {missing}")
    rows = [by_id[b] for b in buoy_order]
    kinds = {(r.samples.dtype.str, r.samples.shape, float(r.sample_rate_hz)) for r in rows}
    if len(kinds) != 1:
        raise IqWireError("frames of the group differ in sample type, length or sample rate")
    return np.stack([r.samples for r in rows])
