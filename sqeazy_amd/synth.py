"""Deterministic synthetic microscopy-like stacks.

Integer-only generator patterned on the reference's "noisy embryo" fixture
(src/cpp/bench/benchmark_fixtures.hpp:85-119, tests/volume_fixtures.hpp:18-88, which is itself
non-deterministic): camera-like bell-shaped noise + an ellipsoid-shell signal.  Voxel (z,y,x) with
linear index i:  r = splitmix64(seed ^ i);  noise = sum of the four low bytes of r (0..1020).
  u16: v = 100 + (noise >> 2) + shell * 6000
  u8 : v = 16 + (noise >> 4) + shell * 120 + (z * 37) % 13
shell = 1 iff |64*(dx^2/(X/4)^2 + dy^2/(Y/4)^2 + dz^2/(0.6 Z)^2) - 64| < 5 with dx = 2x - X etc.
"""
import numpy as np

SEED = 0x5EA2


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _noise(start, count, seed):
    with np.errstate(over="ignore"):
        i = np.arange(start, start + count, dtype=np.uint64)
        r = splitmix64(np.uint64(seed) ^ i)
    b = r.view(np.uint8).reshape(-1, 8)
    return b[:, 0].astype(np.uint32) + b[:, 1] + b[:, 2] + b[:, 3]


def _shell(z0, nz, Z, Y, X):
    """shell mask for frames [z0, z0+nz) of a Z,Y,X volume (integer arithmetic)"""
    dz = (2 * np.arange(z0, z0 + nz, dtype=np.int64) - Z)[:, None, None]
    dy = (2 * np.arange(Y, dtype=np.int64) - Y)[None, :, None]
    dx = (2 * np.arange(X, dtype=np.int64) - X)[None, None, :]
    ax = max(X // 4, 1) ** 2
    ay = max(Y // 4, 1) ** 2
    az = max((6 * Z) // 10, 1) ** 2
    q = (64 * dx * dx) // ax + (64 * dy * dy) // ay + (64 * dz * dz) // az
    return np.abs(q - 64) < 5


def stack(shape, dtype=np.uint16, seed=SEED, frames_per_step=16):
    """returns the synthetic stack of `shape` = (Z, Y, X) as a C-contiguous ndarray"""
    Z, Y, X = (int(s) for s in shape)
    dtype = np.dtype(dtype)
    out = np.empty((Z, Y, X), dtype=dtype)
    per = Y * X
    for z0 in range(0, Z, frames_per_step):
        nz = min(frames_per_step, Z - z0)
        noise = _noise(z0 * per, nz * per, seed).reshape(nz, Y, X)
        sh = _shell(z0, nz, Z, Y, X)
        if dtype == np.uint16:
            v = 100 + (noise >> 2) + sh * 6000
        elif dtype == np.uint8:
            zoff = ((np.arange(z0, z0 + nz) * 37) % 13)[:, None, None]
            v = 16 + (noise >> 4) + sh * 120 + zoff
        else:
            raise TypeError(dtype)
        out[z0:z0 + nz] = v.astype(dtype)
    return out


# ---- the same generator on an HBM-resident torch tensor (bench.py; inputs never touch the host) ----
def _i64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def stack_torch(shape, dtype, device, seed=SEED, z_offset=0, z_total=None, frames_per_step=32):
    """torch twin of `stack`: frames [z_offset, z_offset + shape[0]) of a (z_total, Y, X) volume.
    Bit-identical to the numpy generator (checked in tests/)."""
    import torch
    Z, Y, X = (int(s) for s in shape)
    ZT = int(z_total) if z_total is not None else Z
    tdtype = {np.dtype(np.uint16): torch.uint16, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
    out = torch.empty((Z, Y, X), dtype=tdtype, device=device)
    per = Y * X
    m34, m37, m33 = (1 << 34) - 1, (1 << 37) - 1, (1 << 33) - 1
    c0, c1, c2 = _i64(0x9E3779B97F4A7C15), _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)
    ax, ay, az = max(X // 4, 1) ** 2, max(Y // 4, 1) ** 2, max((6 * ZT) // 10, 1) ** 2
    dy = (2 * torch.arange(Y, dtype=torch.int64, device=device) - Y)[None, :, None]
    dx = (2 * torch.arange(X, dtype=torch.int64, device=device) - X)[None, None, :]
    qyx = torch.div(64 * dx * dx, ax, rounding_mode="floor") + torch.div(64 * dy * dy, ay, rounding_mode="floor")
    for z0 in range(0, Z, frames_per_step):
        nz = min(frames_per_step, Z - z0)
        zg = z0 + z_offset
        i = torch.arange(zg * per, (zg + nz) * per, dtype=torch.int64, device=device)
        x = (i ^ seed) + c0
        z = (x ^ ((x >> 30) & m34)) * c1
        z = (z ^ ((z >> 27) & m37)) * c2
        r = z ^ ((z >> 31) & m33)
        noise = (r & 0xFF) + ((r >> 8) & 0xFF) + ((r >> 16) & 0xFF) + ((r >> 24) & 0xFF)
        noise = noise.reshape(nz, Y, X)
        zz = torch.arange(zg, zg + nz, dtype=torch.int64, device=device)
        dz = (2 * zz - ZT)[:, None, None]
        q = qyx + torch.div(64 * dz * dz, az, rounding_mode="floor")
        sh = ((q - 64).abs() < 5).to(torch.int64)
        if tdtype == torch.uint16:
            v = 100 + (noise >> 2) + sh * 6000
        else:
            v = 16 + (noise >> 4) + sh * 120 + ((zz * 37) % 13)[:, None, None]
        out[z0:z0 + nz] = v.to(tdtype)
    return out
