"""Synthetic scenes for the scan-to-map path.

Procedural, seeded (default 20250204) street scene: a ground plane at z = -1.73 m
(HDL-64 mount height), two street facades and >= 40 axis-aligned boxes, all analytic
so lidar rays intersect in closed form.  Produces

* a local surf MAP in the world frame: surfaces sampled on a jittered lattice with
  N(0, 0.02 m) noise, voxel-thinned to one centroid per voxel (what the reference's
  VoxelGrid over laserCloudSurfFromMap does, src/mapOptmization.cpp:1037-1038) and cut
  to an exact point count in a seeded random order, and
* lidar SCANS in the lidar frame: analytic ray casts from a ground-truth sensor pose
  with N(0, 0.02 m) range noise (Velodyne-64 / Ouster-128 / dense rosette patterns),

plus the ground-truth pose and the perturbed initial guess.  This is workload data for
tests/ and bench.py; it is not part of the registration path itself.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

SEED = 20250204
GROUND_Z = -1.73
POINT_STRIDE = 32  # sizeof(pcl::PointXYZI)

# ground-truth sensor pose and initial-guess offset, [roll,pitch,yaw,x,y,z]
POSE_GT = np.array([0.01, -0.02, 0.3, 1.5, -0.7, 0.1], dtype=np.float32)
POSE_DELTA = np.array([math.radians(0.5), math.radians(-0.3), math.radians(1.0), 0.10, -0.08, 0.05],
                      dtype=np.float32)


def rotation_rpy(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """R = Rz(yaw) Ry(pitch) Rx(roll) (pcl::getTransformation convention), float64."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, sy * sr + cy * sp * cr],
        [sy * cp, cy * cr + sy * sp * sr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]], dtype=np.float64)


@dataclass
class Scene:
    half: float                     # scene spans |x|,|y| <= half
    facades: np.ndarray             # (F,4): y0, x_min, x_max, height
    boxes: np.ndarray               # (B,5): cx, cy, hx, hy, height (sitting on the ground)
    seed: int = SEED
    extra: dict = field(default_factory=dict)


def make_scene(seed: int = SEED, half: float = 70.0, n_boxes: int = 48) -> Scene:
    rng = np.random.default_rng(seed)
    facades = np.array([[18.0 / 70.0 * half, -half, half, 12.0], [-22.0 / 70.0 * half, -half, half, 9.0]],
                       dtype=np.float64)
    boxes = []
    tries = 0
    while len(boxes) < n_boxes and tries < 100000:
        tries += 1
        hx, hy = rng.uniform(1.0, 9.0), rng.uniform(1.0, 7.0)
        h = rng.uniform(1.2, 14.0) if rng.random() < 0.7 else rng.uniform(1.2, 2.2)   # buildings / cars
        cx, cy = rng.uniform(-half + hx, half - hx), rng.uniform(-half + hy, half - hy)
        if abs(cx - POSE_GT[3]) < hx + 6.0 and abs(cy - POSE_GT[4]) < hy + 6.0:
            continue        # keep the sensor's path clear
        boxes.append((cx, cy, hx, hy, h))
    return Scene(half=half, facades=facades, boxes=np.array(boxes, dtype=np.float64), seed=seed)


# --------------------------------------------------------------------------- map
def _lattice(rng, u0, u1, v0, v1, pitch):
    nu, nv = max(1, int(math.ceil((u1 - u0) / pitch))), max(1, int(math.ceil((v1 - v0) / pitch)))
    uu = u0 + (np.arange(nu) + 0.5) * (u1 - u0) / nu
    vv = v0 + (np.arange(nv) + 0.5) * (v1 - v0) / nv
    U, V = np.meshgrid(uu, vv, indexing="ij")
    U = U.ravel() + rng.uniform(-0.45, 0.45, U.size) * pitch
    V = V.ravel() + rng.uniform(-0.45, 0.45, V.size) * pitch
    return U, V


def _surface_samples(scene: Scene, rng, pitch: float) -> np.ndarray:
    parts = []
    h = scene.half
    U, V = _lattice(rng, -h, h, -h, h, pitch)
    parts.append(np.stack([U, V, np.full_like(U, GROUND_Z)], 1))
    for y0, x0, x1, ht in scene.facades:
        U, V = _lattice(rng, x0, x1, GROUND_Z, GROUND_Z + ht, pitch)
        parts.append(np.stack([U, np.full_like(U, y0), V], 1))
    for cx, cy, hx, hy, ht in scene.boxes:
        z0, z1 = GROUND_Z, GROUND_Z + ht
        for sx in (-1.0, 1.0):
            U, V = _lattice(rng, cy - hy, cy + hy, z0, z1, pitch)
            parts.append(np.stack([np.full_like(U, cx + sx * hx), U, V], 1))
        for sy in (-1.0, 1.0):
            U, V = _lattice(rng, cx - hx, cx + hx, z0, z1, pitch)
            parts.append(np.stack([U, np.full_like(U, cy + sy * hy), V], 1))
        U, V = _lattice(rng, cx - hx, cx + hx, cy - hy, cy + hy, pitch)
        parts.append(np.stack([U, V, np.full_like(U, z1)], 1))
    return np.concatenate(parts, 0)


def voxel_thin(points: np.ndarray, leaf: float) -> np.ndarray:
    """One centroid per occupied voxel (numpy statement of a centroid voxel grid)."""
    key = np.floor(points / leaf).astype(np.int64)
    key -= key.min(0)
    dims = key.max(0) + 1
    lin = (key[:, 2] * dims[1] + key[:, 1]) * dims[0] + key[:, 0]
    uniq, inv = np.unique(lin, return_inverse=True)
    cnt = np.bincount(inv, minlength=uniq.size).astype(np.float64)
    out = np.empty((uniq.size, 3), dtype=np.float64)
    for d in range(3):
        out[:, d] = np.bincount(inv, weights=points[:, d], minlength=uniq.size) / cnt
    return out


def make_map(scene: Scene, n_points: int, leaf: float = 0.5, seed: int | None = None,
             noise: float = 0.02) -> np.ndarray:
    """Exactly n_points map points (float32, world frame), order carries no spatial favour."""
    seed = scene.seed if seed is None else seed
    rng = np.random.default_rng(seed + 1)
    pitch = leaf * 0.45
    pts = _surface_samples(scene, rng, pitch)
    pts = pts + rng.normal(0.0, noise, pts.shape)
    thin = voxel_thin(pts, leaf)
    if thin.shape[0] < n_points:
        raise ValueError(f"scene too small: {thin.shape[0]} voxels < {n_points} requested "
                         f"(raise n_boxes/half or lower leaf)")
    perm = rng.permutation(thin.shape[0])[:n_points]
    return np.ascontiguousarray(thin[perm].astype(np.float32))


# --------------------------------------------------------------------------- scans
def _cast(scene: Scene, origin: np.ndarray, dirs: np.ndarray, max_range: float) -> np.ndarray:
    """Nearest hit distance per ray (inf = miss). dirs are unit vectors in the world frame."""
    n = dirs.shape[0]
    t_best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground
        t = (GROUND_Z - origin[2]) / dirs[:, 2]
        p = origin[None, :2] + t[:, None] * dirs[:, :2]
        ok = (t > 0) & (np.abs(p[:, 0]) <= scene.half) & (np.abs(p[:, 1]) <= scene.half)
        t_best = np.where(ok & (t < t_best), t, t_best)
        # facades (planes y = y0)
        for y0, x0, x1, ht in scene.facades:
            t = (y0 - origin[1]) / dirs[:, 1]
            px = origin[0] + t * dirs[:, 0]
            pz = origin[2] + t * dirs[:, 2]
            ok = (t > 0) & (px >= x0) & (px <= x1) & (pz >= GROUND_Z) & (pz <= GROUND_Z + ht)
            t_best = np.where(ok & (t < t_best), t, t_best)
        # boxes (slab test)
        inv = 1.0 / dirs
        for cx, cy, hx, hy, ht in scene.boxes:
            lo = np.array([cx - hx, cy - hy, GROUND_Z])
            hi = np.array([cx + hx, cy + hy, GROUND_Z + ht])
            t0 = (lo[None, :] - origin[None, :]) * inv
            t1 = (hi[None, :] - origin[None, :]) * inv
            tmin = np.nanmax(np.minimum(t0, t1), axis=1)
            tmax = np.nanmin(np.maximum(t0, t1), axis=1)
            ok = (tmax >= tmin) & (tmin > 0)
            t_best = np.where(ok & (tmin < t_best), tmin, t_best)
    t_best[t_best > max_range] = np.inf
    return t_best


def _pattern(sensor: str, n_az: int, rng) -> np.ndarray:
    """Unit ray directions in the lidar frame, acquisition order (azimuth-major)."""
    if sensor == "velodyne64":
        elev = np.radians(np.linspace(-24.8, 2.0, 64))
    elif sensor == "ouster128":
        elev = np.radians(np.linspace(-22.5, 22.5, 128))
    elif sensor == "dense":
        # seeded non-repeating rosette: golden-angle azimuths, elevations from a sinusoid mix
        k = np.arange(n_az, dtype=np.float64)
        az = (k * 2.399963229728653) % (2 * math.pi)
        el = np.radians(-25.0 + 32.0 * (0.5 + 0.5 * np.sin(k * 0.61803398875 * 0.37 + rng.uniform(0, 6.28))))
        ce = np.cos(el)
        return np.stack([ce * np.cos(az), ce * np.sin(az), np.sin(el)], 1)
    else:
        raise ValueError(sensor)
    az = (np.arange(n_az) + rng.uniform(0, 1)) * (2 * math.pi / n_az)
    AZ, EL = np.meshgrid(az, elev, indexing="ij")     # azimuth-major
    AZ, EL = AZ.ravel(), EL.ravel()
    ce = np.cos(EL)
    return np.stack([ce * np.cos(AZ), ce * np.sin(AZ), np.sin(EL)], 1)


def make_scan(scene: Scene, pose_gt: np.ndarray, sensor: str, n_points: int, seed: int | None = None,
              noise: float = 0.02, max_range: float = 100.0) -> np.ndarray:
    """Exactly n_points returns (float32, lidar frame), misses dropped, acquisition order."""
    seed = scene.seed if seed is None else seed
    rng = np.random.default_rng(seed + 2)
    R = rotation_rpy(*[float(v) for v in pose_gt[:3]])
    origin = pose_gt[3:].astype(np.float64)
    n_rings = {"velodyne64": 64, "ouster128": 128, "dense": 1}[sensor]
    n_az = int(math.ceil(n_points / n_rings * 1.08))
    for _ in range(12):
        d_l = _pattern(sensor, n_az, rng)
        t = _cast(scene, origin, d_l @ R.T, max_range)
        hit = np.isfinite(t) & (t > 1.0)          # lidarMinRange 1.0 (include/utility.h:208)
        if int(hit.sum()) >= n_points:
            break
        n_az = int(n_az * 1.15) + 1
    else:
        raise ValueError("could not reach the requested number of returns")
    d_l, t = d_l[hit][:n_points], t[hit][:n_points]
    t = t + rng.normal(0.0, noise, t.shape)
    return np.ascontiguousarray((d_l * t[:, None]).astype(np.float32))


# --------------------------------------------------------------------------- configs
def to_xyzi(xyz: np.ndarray) -> np.ndarray:
    """(n,3) float32 -> (n,8) float32 records laid out like pcl::PointXYZI (stride 32 B)."""
    out = np.zeros((xyz.shape[0], 8), dtype=np.float32)
    out[:, :3] = xyz
    out[:, 3] = 1.0
    return out


CONFIGS = {
    # name: (sensor, n_q, n_m, leaf, scene half, n_boxes)
    "tiny":      ("velodyne64", 2000, 6000, 0.5, 20.0, 6),      # CPU test size
    "small":     ("velodyne64", 12000, 30000, 0.5, 35.0, 14),   # parity size (oracle: seconds)
    "kitti64":   ("velodyne64", 120000, 200000, 0.5, 70.0, 92),     # BASELINE configs[0]/[1]
    "ouster128": ("ouster128", 262144, 500000, 0.5, 110.0, 250),    # configs[2]
    "dense1m":   ("dense", 300000, 1000000, 0.3, 100.0, 140),         # configs[4]
}


def pose_init_from(pose_gt: np.ndarray) -> np.ndarray:
    return (pose_gt.astype(np.float32) + POSE_DELTA).astype(np.float32)


def make_config(name: str, seed: int = SEED, scan_index: int = 0):
    """Returns dict(map, scan, pose_gt, pose_init, meta). scan_index > 0 gives the further
    seeded poses along a 10 m path on the same map (BASELINE configs[3])."""
    sensor, n_q, n_m, leaf, half, n_boxes = CONFIGS[name]
    scene = make_scene(seed, half=half, n_boxes=n_boxes)
    pose_gt = POSE_GT.copy()
    if scan_index:
        rng = np.random.default_rng(seed + 100 + scan_index)
        pose_gt[3] += 10.0 * scan_index / 7.0
        pose_gt[4] += float(rng.uniform(-0.3, 0.3))
        pose_gt[2] += float(rng.uniform(-0.05, 0.05))
    m = make_map(scene, n_m, leaf=leaf, seed=seed)
    s = make_scan(scene, pose_gt, sensor, n_q, seed=seed + 10 * scan_index)
    return dict(map=m, scan=s, pose_gt=pose_gt, pose_init=pose_init_from(pose_gt),
                meta=dict(name=name, sensor=sensor, n_q=n_q, n_m=n_m, leaf=leaf, seed=seed,
                          scan_index=scan_index))


def move_config(cfg, yaw: float, t):
    """The same scene somewhere else in the odometry frame: map points and poses under the rigid motion
    p -> Rz(yaw) p + t (the reference's transformTobeMapped and local map live in the odometry / world frame and grow
    without bound, src/mapOptmization.cpp:134,1273-1278); the scan stays in the lidar frame.  fp64 arithmetic, rounded
    to fp32 once."""
    c, s = np.cos(yaw), np.sin(yaw)
    Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    t = np.asarray(t, np.float64)
    out = dict(cfg)
    out["map"] = (cfg["map"].astype(np.float64) @ Rz.T + t).astype(np.float32)
    for key in ("pose_gt", "pose_init"):
        p = cfg[key].astype(np.float64)
        q = p.copy()
        q[2] = (p[2] + yaw + np.pi) % (2.0 * np.pi) - np.pi          # Rz(yaw) Rz(y) Ry(p) Rx(r): the yaw angles add
        q[3:] = Rz @ p[3:] + t
        out[key] = q.astype(np.float32)
    out["meta"] = dict(cfg["meta"], moved=(float(yaw), [float(v) for v in t]))
    return out
