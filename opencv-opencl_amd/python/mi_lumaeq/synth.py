"""Seeded synthetic NV12 frames: counter-based SplitMix64 hash, no rand().

Y distributions:  D1 uniform bytes | D2 "natural low-contrast" (gradient + triangular noise,
clamped to [16,200], ~60 populated bins) | D3 constant 128 | D4 two-valued 16/235 checkerboard |
D5 full horizontal ramp.  UV plane: uniform bytes from seed ^ 0xA5A5 (so passthrough vs 128-fill
is checkable).  seed = 0x5EED0000 + frame_index.
"""
from __future__ import annotations

import numpy as np

DISTS = ("D1", "D2", "D3", "D4", "D5")
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """SplitMix64 finaliser of (seed + (idx+1)*golden); idx uint64 array -> uint64 array."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def random_bytes(n: int, seed: int) -> np.ndarray:
    words = splitmix64(np.arange((n + 7) // 8, dtype=np.uint64), seed)
    return words.view(np.uint8)[:n].copy()


def frame_seed(frame_index: int) -> int:
    return 0x5EED0000 + int(frame_index)


def y_plane(width: int, height: int, dist: str = "D1", frame_index: int = 0) -> np.ndarray:
    seed = frame_seed(frame_index)
    n = width * height
    if dist == "D1":
        return random_bytes(n, seed).reshape(height, width)
    if dist == "D2":
        r = random_bytes(2 * n, seed).astype(np.int32)
        tri = (r[:n] % 25) + (r[n:] % 25) - 24                         # triangular noise in [-24, 24]
        xs = np.arange(width, dtype=np.int32)[None, :]
        ys = np.arange(height, dtype=np.int32)[:, None]
        grad = (xs * 128 // max(width, 1) + ys * 64 // max(height, 1)) // 4   # 0.25 * gradient(x,y)
        return np.clip(96 + grad + tri.reshape(height, width), 16, 200).astype(np.uint8)
    if dist == "D3":
        return np.full((height, width), 128, np.uint8)
    if dist == "D4":
        xs = np.arange(width)[None, :] // 8
        ys = np.arange(height)[:, None] // 8
        return np.where((xs + ys + frame_index) % 2 == 0, 16, 235).astype(np.uint8)
    if dist == "D5":
        ramp = (np.arange(width, dtype=np.int64) * 256 // max(width, 1)).astype(np.uint8)
        return np.broadcast_to(ramp[None, :], (height, width)).copy()
    raise ValueError(f"unknown distribution {dist!r}")


def photo_like(width: int, height: int, seed: int = 1, channels: int = 1) -> np.ndarray:
    """A synthetic stand-in for a photograph (`hun.png` row: regenerate, do not ship the reference's image):
    piecewise-smooth gradients, large flat regions (some saturated at 0 / 255), soft blobs and fine texture in part of the
    scene -- hot histogram bins, runs of equal neighbours, smooth neighbourhoods, i.e. what the noise distributions D1..D5
    do not produce.  Integer arithmetic on a counter-based hash only, so the bytes are the same on every platform.
    Returns HxW (channels == 1) or HxWx3 uint8 (each channel its own gradients over the same geometry)."""
    w, h = int(width), int(height)
    par = splitmix64(np.arange(512, dtype=np.uint64), seed).astype(np.uint64)
    pi = [0]

    def nxt(lo, hi):                                                # next parameter in [lo, hi)
        v = int(par[pi[0] % 512] % np.uint64(max(hi - lo, 1))) + lo
        pi[0] += 1
        return v
    xs = np.arange(w, dtype=np.int64)[None, :]
    ys = np.arange(h, dtype=np.int64)[:, None]
    # geometry shared by all channels: three cutting lines -> up to 8 cells, flat rectangles / ellipses, soft blobs
    lines = [(nxt(-8, 9), nxt(-8, 9), nxt(0, w), nxt(0, h)) for _ in range(3)]
    cell = np.zeros((h, w), np.int64)
    for k, (a, b, cx, cy) in enumerate(lines):
        if a == 0 and b == 0:
            a = 1
        cell |= (((xs - cx) * a + (ys - cy) * b) > 0).astype(np.int64) << k
    rects = [(nxt(0, w), nxt(0, h), nxt(w // 16 + 1, w // 4 + 2), nxt(h // 16 + 1, h // 4 + 2)) for _ in range(6)]
    ells = [(nxt(0, w), nxt(0, h), nxt(w // 20 + 2, w // 6 + 3), nxt(h // 20 + 2, h // 6 + 3)) for _ in range(4)]
    blobs = [(nxt(0, w), nxt(0, h), nxt(max(w, h) // 8 + 2, max(w, h) // 3 + 3)) for _ in range(5)]
    flat_vals = [0, 255, 16, 235, nxt(30, 220), nxt(30, 220), nxt(30, 220), nxt(30, 220), nxt(30, 220), nxt(30, 220)]
    tex = (splitmix64((ys * w + xs).astype(np.uint64).reshape(-1), seed ^ 0x7E57).reshape(h, w) % np.uint64(5)).astype(np.int64) - 2
    planes = []
    for c in range(max(1, channels)):
        img = np.zeros((h, w), np.int64)
        for cid in range(8):                                        # every cell its own gradient plane
            gx, gy, off = nxt(-90, 91), nxt(-90, 91), nxt(50, 200)
            plane = off + (xs - w // 2) * gx // max(w, 1) + (ys - h // 2) * gy // max(h, 1)
            img = np.where(cell == cid, plane, img)
        for (cx, cy, r) in blobs:                                   # soft light / shadow: quadratic falloff, integer
            amp = nxt(-50, 51)
            d2 = (xs - cx) ** 2 + (ys - cy) ** 2
            img = img + np.where(d2 < r * r, amp * (r * r - d2) // (r * r), 0)
        img = img + np.where((cell & 1) == 1, tex, 0)               # fine texture in half of the cells only
        for k, (x0, y0, rw, rh) in enumerate(rects):
            img = np.where((xs >= x0) & (xs < x0 + rw) & (ys >= y0) & (ys < y0 + rh), flat_vals[k] if c == 0 else (flat_vals[k] * (c + 2) // 3) , img)
        for k, (cx, cy, rx, ry) in enumerate(ells):
            inside = ((xs - cx) ** 2) * (ry * ry) + ((ys - cy) ** 2) * (rx * rx) < (rx * rx) * (ry * ry)
            img = np.where(inside, flat_vals[6 + k], img)
        planes.append(np.clip(img, 0, 255).astype(np.uint8))
    return planes[0] if channels == 1 else np.ascontiguousarray(np.stack(planes, axis=-1))


def uv_plane(width: int, height: int, frame_index: int = 0) -> np.ndarray:
    return random_bytes((width * height) // 2, frame_seed(frame_index) ^ 0xA5A5)


def nv12_frame(width: int, height: int, dist: str = "D1", frame_index: int = 0) -> np.ndarray:
    return np.concatenate([y_plane(width, height, dist, frame_index).reshape(-1), uv_plane(width, height, frame_index)])


def nv12_batch(width: int, height: int, n_frames: int, dist: str = "D1", first_index: int = 0) -> np.ndarray:
    fb = width * height + (width * height) // 2
    out = np.empty((n_frames, fb), np.uint8)
    for k in range(n_frames):
        out[k] = nv12_frame(width, height, dist, first_index + k)
    return out


def nv12_batch_torch(width: int, height: int, n_frames: int, dist: str, device, seed: int = 0x5EED0000):
    """Same distributions generated on the GPU with torch's own RNG (bench.py: 64 4K frames would take
    ~10 s through the numpy hash).  Not bit-identical to the numpy generator; parity checks download
    the frames they verify."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ysz, uvsz = width * height, (width * height) // 2
    out = torch.empty((n_frames, ysz + uvsz), dtype=torch.uint8, device=device)
    for k in range(n_frames):
        if dist == "D1":
            y = torch.randint(0, 256, (ysz,), dtype=torch.uint8, device=device, generator=g)
        elif dist == "D2":
            r = torch.randint(0, 25, (2, ysz), dtype=torch.int32, device=device, generator=g)
            tri = (r[0] + r[1] - 24).view(height, width)
            xs = torch.arange(width, dtype=torch.int32, device=device)[None, :]
            ys = torch.arange(height, dtype=torch.int32, device=device)[:, None]
            grad = torch.div(torch.div(xs * 128, max(width, 1), rounding_mode="floor") +
                             torch.div(ys * 64, max(height, 1), rounding_mode="floor"), 4, rounding_mode="floor")
            y = (96 + grad + tri).clamp_(16, 200).to(torch.uint8).view(-1)
        elif dist == "D3":
            y = torch.full((ysz,), 128, dtype=torch.uint8, device=device)
        elif dist == "D4":
            xs = torch.arange(width, device=device)[None, :] // 8
            ys = torch.arange(height, device=device)[:, None] // 8
            y = torch.where((xs + ys + k) % 2 == 0, 16, 235).to(torch.uint8).view(-1)
        elif dist == "D5":
            ramp = (torch.arange(width, device=device, dtype=torch.int64) * 256 // max(width, 1)).to(torch.uint8)
            y = ramp[None, :].expand(height, width).reshape(-1)
        else:
            raise ValueError(dist)
        out[k, :ysz] = y
        out[k, ysz:] = torch.randint(0, 256, (uvsz,), dtype=torch.uint8, device=device, generator=g)
    return out
