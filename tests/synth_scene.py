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
                   max_depth=5.0, seed=0, noise=True, first=0, images=False):
    """Returns a list of dicts {xyz f32[n,3], rgb u8[n,3], kfid u32[n], Twc f32[3,4]}; images=True adds the key frame as
    the images the cloud was made from (depth_grid, rgb_grid on the stride-`step` grid, cam_grid: see stream_keyframe)."""
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
        if images:
            va, ua = np.mgrid[0:d32.shape[0], 0:d32.shape[1]]
            out[-1]["depth_grid"] = np.ascontiguousarray(d32)
            out[-1]["rgb_grid"] = np.ascontiguousarray(np.stack([(ua * 3 + k) & 255, (va * 5 + 2 * k) & 255, (ua + va) & 255],
                                                                axis=-1).astype(np.uint8))
            out[-1]["cam_grid"] = np.ascontiguousarray(np.stack([gx32, gy32], axis=-1).reshape(-1, 2))
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


# ---------------------------------------------------------------------------------------------- a long trajectory
# BASELINE configs[2] is TUM fr3/long_office_household: ~2 500 frames of a hand-held sensor circling a desk island in a
# large office (21.5 m of path, 0.25 m/s, 10 deg/s).  TUM data are not in the tree; this is the stand-in with the same
# shape: a 9.5 x 7.5 x 3 m office, a desk island (boxes, spheres) in the middle, a partition and cabinets along the walls,
# and a camera that walks ONE closed loop of `LOOP` distinct poses around the island, gazing ahead and inwards (6 mm and
# 0.144 deg per key frame, with a hand-held wobble in height and gaze; 36 000 - 73 000 of the 76 800 grid points have a
# depth below 5 m).  Every key frame is different; a stretch of 100 consecutive ones sees the island from a moving side
# (revisits, from new angles) and sweeps 14 deg of fresh walls and furniture.
# Poses beyond LOOP continue on further laps at other heights with other noise (still all distinct).
LOOP = 2500
_OFFICE = (np.array([-4.75, -3.75, -1.5]), np.array([4.75, 3.75, 1.5]))
_BOXES = [  # (lo, hi): desk island, two monitors, partition, cabinets, a shelf
    (np.array([-1.6, -0.8, -1.5]), np.array([1.6, 0.8, -0.75])),
    (np.array([-1.2, -0.1, -0.75]), np.array([-0.5, 0.0, -0.3])),
    (np.array([0.4, -0.1, -0.75]), np.array([1.1, 0.0, -0.3])),
    (np.array([-4.75, 2.6, -1.5]), np.array([-3.6, 2.7, 0.6])),
    (np.array([2.2, -3.75, -1.5]), np.array([4.0, -3.2, 0.5])),
    (np.array([-4.2, -3.75, -1.5]), np.array([-2.9, -3.1, 0.2])),
    (np.array([4.3, -1.5, -1.5]), np.array([4.75, 1.5, 0.9])),
    (np.array([-0.4, 3.2, -1.5]), np.array([2.4, 3.75, -0.2])),
]
_SPHERES = [(np.array([0.0, 0.3, -0.45]), 0.3), (np.array([-1.0, -0.4, -0.55]), 0.2), (np.array([1.2, 0.45, -0.5]), 0.25),
            (np.array([3.9, 2.9, -1.0]), 0.5), (np.array([-4.1, -1.2, -1.1]), 0.4), (np.array([-2.6, 3.1, -0.9]), 0.6)]


def stream_pose(k):
    """Twc (3x4, float64) of key frame k of the long trajectory."""
    lap, th = divmod(k, LOOP)
    th = 2.0 * np.pi * th / LOOP
    pos = np.array([2.8 * np.cos(th), 2.0 * np.sin(th), 0.12 * np.sin(5 * th) + 0.07 * lap - 0.05])
    tgt = np.array([2.4 * np.cos(th + 0.8), 1.7 * np.sin(th + 0.8), -0.55 + 0.15 * np.sin(7 * th + lap)])
    fwd = tgt - pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.concatenate([np.stack([right, down, fwd], axis=1), pos[:, None]], axis=1)


def _stream_depth(T, cam, step):
    us = np.arange(0, cam["width"], step, dtype=np.float32)
    vs = np.arange(0, cam["height"], step, dtype=np.float32)
    GX, GY = np.meshgrid((us - np.float32(cam["cx"])) / np.float32(cam["fx"]),
                         (vs - np.float32(cam["cy"])) / np.float32(cam["fy"]))
    R, o = T[:, :3].astype(np.float32), T[:, 3].astype(np.float32)
    d = [GX * R[i, 0] + GY * R[i, 1] + R[i, 2] for i in range(3)]            # world direction per pixel (z-depth param)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = [np.float32(1.0) / di for di in d]
        lo, hi = _OFFICE
        depth = np.minimum(np.minimum(np.maximum((lo[0] - o[0]) * inv[0], (hi[0] - o[0]) * inv[0]),
                                      np.maximum((lo[1] - o[1]) * inv[1], (hi[1] - o[1]) * inv[1])),
                           np.maximum((lo[2] - o[2]) * inv[2], (hi[2] - o[2]) * inv[2]))
        for lo, hi in _BOXES:
            tn, tf = None, None
            for i in range(3):
                a, b = (np.float32(lo[i]) - o[i]) * inv[i], (np.float32(hi[i]) - o[i]) * inv[i]
                n_, f_ = np.minimum(a, b), np.maximum(a, b)
                tn = n_ if tn is None else np.maximum(tn, n_)
                tf = f_ if tf is None else np.minimum(tf, f_)
            depth = np.where((tf >= tn) & (tn > 1e-4), np.minimum(depth, tn), depth)
        a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
        for c, r in _SPHERES:
            oc = (o - c.astype(np.float32))
            b = 2.0 * (d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2])
            disc = b * b - 4.0 * a * np.float32(oc @ oc - r * r)
            ts = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2.0 * a), np.inf)
            depth = np.where(ts > 1e-4, np.minimum(depth, ts), depth)
    return GX, GY, depth.astype(np.float32)


def stream_keyframe(k, cam=None, step=2, min_depth=0.1, max_depth=5.0, seed=0, images=False):
    """Key frame k of the long trajectory: {xyz f32[n,3], rgb u8[n,3], kfid u32[n], Twc f32[3,4]} (deterministic in k)."""
    cam = cam or TUM1
    T = stream_pose(k)
    GX, GY, depth = _stream_depth(T, cam, step)
    rng = np.random.default_rng([seed, k])
    sigma = np.float32(0.0012) + np.float32(0.0019) * (depth - np.float32(0.4)) ** 2
    d32 = (depth + rng.standard_normal(depth.shape, dtype=np.float32) * sigma).astype(np.float32)
    d32[rng.random(d32.shape, dtype=np.float32) < 0.01] = 0.0                 # no-return pixels
    keep = (d32 > np.float32(min_depth)) & (d32 < np.float32(max_depth))
    xyz = np.stack([GX * d32, GY * d32, d32], axis=-1)[keep]
    vv, uu = np.nonzero(keep)
    rgb = np.stack([(uu * 3 + k) & 255, (vv * 5 + 2 * k) & 255, (uu + vv + (k >> 3)) & 255], axis=-1)
    out = dict(xyz=np.ascontiguousarray(xyz, dtype=np.float32), rgb=np.ascontiguousarray(rgb, dtype=np.uint8),
               kfid=np.full(xyz.shape[0], k, dtype=np.uint32), Twc=np.ascontiguousarray(T.astype(np.float32)))
    if images:
        # the same key frame as the IMAGES PointCloudMapping starts from (the depth-image entry point of the TSDF,
        # plvs_hip_tsdf_chisel_integrate_depth_batch_dev): depth and colour on the stride-`step` grid (the pixels in between
        # are never read: GeneratePointCloudInCameraFrameBGRA visits m, n = 0, step, 2 step, ...) and the grid table the
        # points above were made with (matCamGridPoints_ of this synthetic camera)
        vv, uu = np.mgrid[0:d32.shape[0], 0:d32.shape[1]]
        out["depth_grid"] = np.ascontiguousarray(d32)
        out["rgb_grid"] = np.ascontiguousarray(np.stack([(uu * 3 + k) & 255, (vv * 5 + 2 * k) & 255, (uu + vv + (k >> 3)) & 255],
                                                        axis=-1).astype(np.uint8))
        out["cam_grid"] = np.ascontiguousarray(np.stack([GX, GY], axis=-1).reshape(-1, 2).astype(np.float32))
    return out


def make_stream_keyframes(n_keyframes, first=0, threads=16, **kw):
    """Key frames first .. first + n_keyframes - 1 of the long trajectory (numpy on `threads` host threads)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, threads)) as ex:
        return list(ex.map(lambda k: stream_keyframe(k, **kw), range(first, first + n_keyframes)))
