"""The viewer socket PEGASUS's loops poll when GUI=True (/root/reference/pegasus.py:85,249-279;
/root/reference/src/gs/gs_object_rotation.py:68-84): ``init`` opens a listening socket, ``try_connect`` accepts a viewer
without blocking, ``receive`` reads one camera request, ``send`` answers with the rendered image.

Wire format (what the callers' use of the return values and the SIBR remote viewer imply; [UPSTREAM-KNOWLEDGE], not in
/root/reference): every message is a 4-byte little-endian length followed by that many bytes.  A request is JSON with
``resolution_x``, ``resolution_y``, ``train``, ``fov_y``, ``fov_x``, ``z_near``, ``z_far``, ``shs_python``,
``rot_scale_python``, ``keep_alive``, ``scaling_modifier`` and the two 4x4 matrices ``view_matrix`` /
``view_projection_matrix`` as 16 row-major floats in the viewer's convention (y and z axes flipped against the
rasterizer's).  The answer is the raw uint8 RGB image (nothing when no camera was requested) followed by a
length-prefixed ASCII string the viewer checks (the dataset path).

Out of the measured hot path; here so that ``from gaussian_renderer import network_gui`` and
the GUI branch of the reference's loops work instead of raising."""
from __future__ import annotations

import json
import socket
import struct

import torch

from .cameras import MiniCam

host = "127.0.0.1"
port = 6009
conn = None            # the connected viewer (the callers test and reset this attribute)
addr = None
device = "cuda"        # where the request's matrices are put (the rasterizer reads them on the device)
_listener = None


def init(wish_host, wish_port):
    """Listen on (wish_host, wish_port); accepting is non-blocking (``try_connect`` is polled once per frame)."""
    global host, port, _listener
    host, port = wish_host, int(wish_port)
    if _listener is not None:
        _listener.close()
    _listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    _listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    _listener.bind((host, port))
    _listener.listen()
    _listener.settimeout(0)
    port = _listener.getsockname()[1]


def try_connect():
    global conn, addr
    if _listener is None:
        return
    try:
        conn, addr = _listener.accept()
    except (BlockingIOError, socket.timeout, OSError):
        return
    conn.settimeout(None)


def _read_exact(n: int) -> bytes:
    chunks, left = [], n
    while left:
        part = conn.recv(left)
        if not part:
            raise ConnectionError("viewer closed the connection")
        chunks.append(part)
        left -= len(part)
    return b"".join(chunks)


def _read_message() -> dict:
    (length,) = struct.unpack("<i", _read_exact(4))
    return json.loads(_read_exact(length).decode("utf-8"))


def _viewer_matrix(values, flip_columns) -> torch.Tensor:
    m = torch.tensor(values, dtype=torch.float32).reshape(4, 4)
    for c in flip_columns:                      # the viewer's y (and z) axes point the other way
        m[:, c] = -m[:, c]
    return m.to(device)


def receive():
    """One request -> (camera or None, do_training, convert_SHs_python, compute_cov3D_python, keep_alive, scaling_modifier)."""
    msg = _read_message()
    width, height = int(msg["resolution_x"]), int(msg["resolution_y"])
    cam = None
    if width != 0 and height != 0:
        cam = MiniCam(width, height, msg["fov_y"], msg["fov_x"], msg["z_near"], msg["z_far"],
                      _viewer_matrix(msg["view_matrix"], (1, 2)), _viewer_matrix(msg["view_projection_matrix"], (1,)))
    return (cam, bool(msg["train"]), bool(msg["shs_python"]), bool(msg["rot_scale_python"]), bool(msg["keep_alive"]),
            msg["scaling_modifier"])


def send(message_bytes, verify):
    if message_bytes is not None:
        conn.sendall(message_bytes)
    text = str(verify).encode("ascii", errors="replace")
    conn.sendall(struct.pack("<i", len(text)) + text)
