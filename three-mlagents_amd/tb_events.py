"""TensorBoard scalar logs without the tensorboard package: a minimal writer of the `events.out.tfevents.*` record format, so the
`rollout/*`, `train/*` and `time/*` scalars SB3's logger sends to `tensorboard_log` (reference: `backend/mlagents/training.py:152-161`,
`tensorboard_log=str(tb_dir)`) appear where a TensorBoard pointed at the run directory expects them.

Format (public: TFRecord framing + the Event / Summary protobuf schema):
  record  = u64le length | u32le masked_crc32c(length bytes) | payload | u32le masked_crc32c(payload)
  payload = Event { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary { 1: Value { 1: string tag, 2: float simple_value } } }
"""
from __future__ import annotations

import os
import socket
import struct
import time

_POLY = 0x82F63B78  # CRC-32C (Castagnoli), reflected
_TABLE = []
for _n in range(256):
    _c = _n
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(number: int, payload: bytes) -> bytes:
    return _varint((number << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int, scalars: dict[str, float] | None = None, file_version: str | None = None) -> bytes:
    ev = b"\x09" + struct.pack("<d", wall_time) + b"\x10" + _varint(step)
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b"".join(_field_bytes(1, _field_bytes(1, tag.encode()) + b"\x15" + struct.pack("<f", float(v))) for tag, v in scalars.items())
        ev += _field_bytes(5, summary)
    return ev


def frame(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


class EventWriter:
    """Appends scalar events to `<logdir>/events.out.tfevents.<time>.<host>`; one file per writer."""

    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}")
        with open(self.path, "wb") as f:
            f.write(frame(encode_event(time.time(), 0, file_version="brain.Event:2")))

    def add_scalars(self, scalars: dict[str, float], step: int, wall_time: float | None = None) -> None:
        finite = {k: float(v) for k, v in scalars.items() if isinstance(v, (int, float)) and v == v}
        if not finite:
            return
        with open(self.path, "ab") as f:
            f.write(frame(encode_event(time.time() if wall_time is None else wall_time, int(step), finite)))


def read_scalars(path: str) -> list[tuple[int, dict[str, float]]]:
    """Parse an event file back (checks both CRCs of every record): [(step, {tag: value})].  Used by the tests and for inspection."""

    def fields(buf: bytes):
        i = 0
        while i < len(buf):
            key, shift = 0, 0
            while True:
                b = buf[i]
                i += 1
                key |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            num, wire = key >> 3, key & 7
            if wire == 0:
                val, shift = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    val |= (b & 0x7F) << shift
                    shift += 7
                    if not b & 0x80:
                        break
                yield num, val
            elif wire == 1:
                yield num, buf[i:i + 8]
                i += 8
            elif wire == 5:
                yield num, buf[i:i + 4]
                i += 4
            else:
                n, shift = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    n |= (b & 0x7F) << shift
                    shift += 7
                    if not b & 0x80:
                        break
                yield num, buf[i:i + n]
                i += n

    out = []
    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[pos + 8:pos + 12])[0] != masked_crc(head):
            raise ValueError("length CRC mismatch")
        payload = data[pos + 12:pos + 12 + n]
        if struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] != masked_crc(payload):
            raise ValueError("payload CRC mismatch")
        pos += 16 + n
        step, scalars = 0, {}
        for num, val in fields(payload):
            if num == 2:
                step = val
            elif num == 5:
                for _, value in fields(val):
                    tag, x = None, None
                    for vn, vv in fields(value):
                        if vn == 1:
                            tag = vv.decode()
                        elif vn == 2:
                            (x,) = struct.unpack("<f", vv)
                    if tag is not None:
                        scalars[tag] = x
        if scalars:
            out.append((step, scalars))
    return out
