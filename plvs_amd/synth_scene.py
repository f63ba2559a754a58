"""Seeded synthetic RGB-D input for the TSDF path: an
axis-aligned room with three spheres, camera on a circle, Kinect-like depth
noise.  Produces exactly what PointCloudMapping hands to
PointCloudMap::InsertData — a camera-frame cloud on the stride-2 pixel grid
(GeneratePointCloudInCameraFrameBGRA, src/PointCloudMapping.cc:929-1030: p =
(gx*d, gy*d, d), kept when minDepth < d < maxDepth) plus the pose Twc.

Input generation only: numpy on the host, no part of the measured path.
"""
import numpy as np

TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, width=640, height=480)
KITTI = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, width=1241, height=376)


def _render_depth(R, t, cam, room, spheres, step):
    """z-depth image on the stride-`step` grid for camera pose (R, t) = Twc."""
    us = np.arange(0, cam["width"], step, dtype=np.float64)
    vs = np.arange(0, cam["height"], step, dtype=np.float64)
    gx = (us - cam["cx"]) / cam["fx"]
    gy = (vs - cam["cy"]) / cam["fy"]
    GX, GY = np.meshgrid(gx, gy)
    d_cam = np.stack([GX, GY, np.ones_like(GX)], axis=-1)          # z = 1 -> ray param = z-depth
    d_w = d_cam @ R.T
    o = t[None, None, :]
    lo, hi = room
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (lo[None, None, :] - o) / d_w
        t2 = (hi[None, None, :] - o) / d_w
    texit = np.where(d_w > 0, t2, t1)
    texit = np.where(d_w == 0, np.inf, texit)
    depth = texit.min(axis=-1)
    for c, r in spheres:
        oc = o - c[None, None, :]
        a = (d_w * d_w).sum(-1)
        b = 2.0 * (d_w * oc).sum(-1)
        cc = (oc * oc).sum(-1) - r * r
        disc = b * b - 4 * a * cc
        ts = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        ts = np.where(ts > 1e-6, ts, np.inf)
        depth = np.minimum(depth, ts)
    return GX, GY, depth


def make_keyframes(n_keyframes=100, cam=None, room_size=(6.0, 4.0, 3.0), step=2, min_depth=0.1,
                   max_depth=5.0, seed=0, noise=True, first=0):
    """Returns a list of dicts {xyz f32[n,3], rgb u8[n,3], kfid u32[n], Twc f32[3,4]}."""
    cam = cam or TUM1
    rng = np.random.default_rng(seed)
    half = np.array(room_size, dtype=np.float64) / 2.0
    room = (-half, half)
    # three spheres r = 0.4 m, all >= 0.9 m from the camera path (an RGB-D sensor of
    # the Kinect/Xtion class has no returns below ~0.5 m; TUM fr1/fr3 depth is 0.5-4 m)
    spheres = [(np.array([2.3, 0.5, 0.2]), 0.4), (np.array([-2.3, -0.4, -0.3]), 0.4),
               (np.array([-2.0, 1.3, 0.4]), 0.4)]
    out = []
    for k in range(first, first + n_keyframes):
        yaw = np.deg2rad(3.6 * k)
        # camera on a circle of radius 1 m in the x-y plane, looking outwards;
        # camera axes: z forward, x right, y down (world z is up)
        pos = np.array([np.cos(yaw), np.sin(yaw), 0.1 * np.sin(2 * yaw)])
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=1)                    # columns = camera axes in world
        GX, GY, depth = _render_depth(R, pos, cam, room, spheres, step)
        if noise:
            sigma = 0.0012 + 0.0019 * (depth - 0.4) ** 2
            depth = depth + rng.standard_normal(depth.shape) * sigma
        d32 = depth.astype(np.float32)
        keep = (d32 > np.float32(min_depth)) & (d32 < np.float32(max_depth))
        gx32, gy32 = GX.astype(np.float32), GY.astype(np.float32)
        xyz = np.stack([gx32 * d32, gy32 * d32, d32], axis=-1)[keep]
        vv, uu = np.nonzero(keep)
        rgb = np.stack([(uu * 3 + k) & 255, (vv * 5 + 2 * k) & 255, (uu + vv) & 255], axis=-1)
        Twc = np.concatenate([R, pos[:, None]], axis=1).astype(np.float32)
        out.append(dict(xyz=np.ascontiguousarray(xyz, dtype=np.float32),
                        rgb=np.ascontiguousarray(rgb, dtype=np.uint8),
                        kfid=np.full(xyz.shape[0], k, dtype=np.uint32),
                        Twc=np.ascontiguousarray(Twc)))
    return out


def make_rgbd_frames(n_frames=2, cam=None, room_size=(6.0, 4.0, 3.0), seed=0, holes=True, first=0):
    """Full-resolution inputs of GeneratePointCloudInCameraFrameBGRA: a list of dicts
    {depth f32[h,w], bgr u8[h,w,3], Twc f32[3,4]}.  `holes` adds what a real sensor produces:
    zero (no return), NaN and out-of-range patches."""
    cam = cam or TUM1
    rng = np.random.default_rng(seed)
    half = np.array(room_size, dtype=np.float64) / 2.0
    spheres = [(np.array([2.3, 0.5, 0.2]), 0.4), (np.array([-2.3, -0.4, -0.3]), 0.4),
               (np.array([-2.0, 1.3, 0.4]), 0.4)]
    out = []
    for k in range(first, first + n_frames):
        yaw = np.deg2rad(3.6 * k)
        pos = np.array([np.cos(yaw), np.sin(yaw), 0.1 * np.sin(2 * yaw)])
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        R = np.stack([right, np.cross(fwd, right), fwd], axis=1)
        _, _, depth = _render_depth(R, pos, cam, (-half, half), spheres, 1)
        depth = depth + rng.standard_normal(depth.shape) * (0.0012 + 0.0019 * (depth - 0.4) ** 2)
        d32 = depth.astype(np.float32)
        h, w = d32.shape
        if holes:
            for _ in range(12):
                y0, x0 = int(rng.integers(0, h - 40)), int(rng.integers(0, w - 40))
                hh, ww = int(rng.integers(3, 40)), int(rng.integers(3, 40))
                d32[y0:y0 + hh, x0:x0 + ww] = (0.0, np.nan, 25.0)[int(rng.integers(0, 3))]
            d32[rng.random(d32.shape) < 0.02] = 0.0
        bgr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        Twc = np.concatenate([R, pos[:, None]], axis=1).astype(np.float32)
        out.append(dict(depth=np.ascontiguousarray(d32), bgr=bgr, Twc=np.ascontiguousarray(Twc)))
    return out
