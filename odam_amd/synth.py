"""Seeded synthetic inputs shaped like a ScanNet scene: camera loop, intrinsics,
multi-view super-quadric problems, RGB frames.  Pure numpy; used by tests, smoke() and bench.py."""
import numpy as np

K_SCANNET = np.array([[577.87, 0.0, 319.5], [0.0, 577.87, 239.5], [0.0, 0.0, 1.0]])
IMG_W, IMG_H = 640, 480


def look_at(eye, target):
    """camera-to-world pose, z forward, y down, world z up"""
    z = target - eye
    z = z / np.linalg.norm(z)
    x = np.cross(z, np.array([0.0, 0.0, 1.0]))
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, eye
    return T


def camera_loop(n, radius=2.6, height=1.3, centre=(0.0, 0.0, 0.5), seed=0):
    rs = np.random.RandomState(seed)
    poses = []
    for i in range(n):
        ang = 2 * np.pi * i / n
        r = radius + 0.2 * np.sin(3 * ang)
        eye = np.array([r * np.cos(ang), r * np.sin(ang), height + 0.15 * np.cos(2 * ang)])
        tgt = np.asarray(centre) + rs.normal(0, 0.02, 3)
        poses.append(look_at(eye, tgt))
    return np.asarray(poses)


def _fexp(x, p):
    return np.sign(x) * np.abs(x) ** p


def sq_surface(centre, yaw, dims, eps, n_eta=60, n_omega=120):
    """dense float64 sampling of a super-quadric surface (ground truth for synthetic boxes);
    semi-axes dims/2, exponents eps"""
    eta = np.linspace(-np.pi / 2, np.pi / 2, n_eta)[:, None]
    om = np.linspace(-np.pi, np.pi, n_omega)[None, :]
    a = np.asarray(dims) / 2.0
    x = a[0] * _fexp(np.cos(eta), eps[0]) * _fexp(np.cos(om), eps[1])
    y = a[1] * _fexp(np.cos(eta), eps[0]) * _fexp(np.sin(om), eps[1])
    z = a[2] * _fexp(np.sin(eta), eps[0]) * np.ones_like(om)
    pts = np.stack([x.ravel(), y.ravel(), z.ravel()], 1)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    return pts @ R.T + np.asarray(centre)


def make_sq_problem(n_views, seed, K=K_SCANNET, img_w=IMG_W, img_h=IMG_H, noise_px=2.0, edge=20):
    """One object seen from n_views cameras on a ring.

    Returns dict: params0-ingredients (translate, angle, dims), class id, P [F,3,4] float64,
    bbox_lines (list of {name: [a, b, -pixel]} like quadric_helper.bbox_to_lines), boxes [F,4].
    """
    rs = np.random.RandomState(seed)
    dims = rs.uniform(0.3, 1.5, 3)
    yaw = rs.uniform(-np.pi, np.pi)
    eps = rs.uniform(0.3, 1.5, 2)
    centre = np.array([rs.uniform(-0.5, 0.5), rs.uniform(-0.5, 0.5), dims[2] / 2])
    surf = sq_surface(centre, yaw, dims, eps)
    Ps, lines, boxes = [], [], []
    phase = rs.uniform(0, 2 * np.pi)
    for f in range(n_views):
        ang = phase + 2 * np.pi * f / n_views
        eye = np.array([2.5 * np.cos(ang), 2.5 * np.sin(ang), 1.3 + 0.2 * np.sin(2 * ang)])
        T_wc = look_at(eye, centre)
        P = K @ np.linalg.inv(T_wc)[:3, :]
        uvw = np.c_[surf, np.ones(len(surf))] @ P.T
        uv = uvw[:, :2] / uvw[:, 2:]
        bb = np.array([uv[:, 0].min(), uv[:, 1].min(), uv[:, 0].max(), uv[:, 1].max()])
        bb = bb + rs.normal(0, noise_px, 4)
        d = {}
        for name, v, lim in zip(("x_min", "y_min", "x_max", "y_max"), bb, (img_w, img_h, img_w, img_h)):
            if edge < v < lim - edge:
                d[name] = np.array([1.0, 0.0, -v]) if name[0] == "x" else np.array([0.0, 1.0, -v])
        Ps.append(P)
        lines.append(d)
        boxes.append(bb)
    return {
        "translate": centre + rs.normal(0, 0.05, 3),
        "angle": yaw + rs.normal(0, 0.1),
        "dims": dims * rs.uniform(0.8, 1.2, 3),
        "class_id": int(rs.randint(0, 8)),
        "P": np.asarray(Ps),
        "bbox_lines": lines,
        "boxes": np.asarray(boxes),
        "gt": {"centre": centre, "yaw": yaw, "dims": dims, "eps": eps},
    }


def make_frames(n, h=IMG_H, w=IMG_W, seed=0, sweep=1.0, noise=24):
    """uint8 RGB frames: smooth low-frequency pattern + uniform noise (generator, one frame at a time).
    `sweep`: fraction of a full period the pattern's phase moves over the n frames; `noise`: amplitude of the
    per-pixel noise (a slowly moving, low-noise stream gives consecutive frames similar detections)"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        ph = 2 * np.pi * sweep * i / max(n, 1)
        base = np.stack([127 + 90 * np.sin(xx / 97.0 + ph) * np.cos(yy / 61.0),
                         127 + 90 * np.sin(xx / 53.0 - ph) * np.sin(yy / 89.0 + ph),
                         127 + 90 * np.cos(xx / 71.0 + yy / 43.0 + ph)], -1)
        img = base + rs.randint(-noise, noise + 1, size=(h, w, 3))
        yield np.clip(img, 0, 255).astype(np.uint8)


def make_sequence(n=40, h=256, w=320, seed=0, sweep=0.02, noise=1):
    """A short posed sequence for whole-pipeline tests: frames (slowly moving pattern), intrinsics for this size, and a
    camera that moves a few centimetres per frame along an arc while looking at a fixed point."""
    frames = list(make_frames(n, h, w, seed=seed, sweep=sweep, noise=noise))
    K = np.array([[0.9 * w, 0.0, (w - 1) / 2.0], [0.0, 0.9 * w, (h - 1) / 2.0], [0.0, 0.0, 1.0]])
    T_wcs = []
    for i in range(n):
        ang = 0.35 * i / max(n - 1, 1)
        eye = np.array([3.0 * np.cos(ang), 3.0 * np.sin(ang), 1.3 + 0.05 * np.sin(3 * ang)])
        T_wcs.append(look_at(eye, np.array([0.0, 0.0, 0.6])))
    return {"frames": frames, "K": K, "T_wcs": np.asarray(T_wcs), "img_names": [3 * i + 1 for i in range(n)], "h": h, "w": w}


def make_scene(n_frames, n_objects, seed=0, K=K_SCANNET, img_w=IMG_W, img_h=IMG_H, min_views=3,
               max_views=None, noise_px=2.0):
    """A posed synthetic scene with object tracks in the reference's 82-column layout
    (src/processor.py:98-108): 0 frame_id | 1 class | 2-5 bbox px | 6-8 dims | 9-11 t_wo |
    12 az_wo | 13 score | 14-77 code (-1) | 78-81 projected bbox px.

    Returns dict(img_names, T_wcs [n,4,4], P_cws [n,3,4], tracks: list of [n_obs,82] float64, K).
    """
    rs = np.random.RandomState(seed)
    T_wcs = camera_loop(n_frames, seed=seed)
    P_cws = np.stack([K @ np.linalg.inv(T)[:3, :] for T in T_wcs])
    img_names = [10 * i for i in range(n_frames)]  # frame ids need not be contiguous
    tracks = []
    for o in range(n_objects):
        dims = rs.uniform(0.3, 1.5, 3)
        yaw = rs.uniform(-np.pi, np.pi)
        eps = rs.uniform(0.3, 1.5, 2)
        centre = np.array([rs.uniform(-1.0, 1.0), rs.uniform(-1.0, 1.0), dims[2] / 2])
        cls = int(rs.randint(0, 8))
        surf = sq_surface(centre, yaw, dims, eps, 40, 80)
        hi = max_views or n_frames
        want = int(rs.randint(min_views, max(min_views + 1, min(hi, n_frames) + 1)))
        start = int(rs.randint(0, n_frames))
        step = max(1, n_frames // max(want, 1)) if want < n_frames else 1
        rows = []
        for j in range(want):
            i = (start + j * step) % n_frames
            uvw = np.c_[surf, np.ones(len(surf))] @ P_cws[i].T
            if uvw[:, 2].min() < 0.3:
                continue
            uv = uvw[:, :2] / uvw[:, 2:]
            bb = np.array([uv[:, 0].min(), uv[:, 1].min(), uv[:, 0].max(), uv[:, 1].max()])
            if bb[2] < 0 or bb[0] > img_w or bb[3] < 0 or bb[1] > img_h:
                continue
            bb = np.clip(bb + rs.normal(0, noise_px, 4), [0, 0, 0, 0], [img_w, img_h, img_w, img_h])
            row = -np.ones(82)
            row[0] = img_names[i]
            row[1] = cls
            row[2:6] = bb
            row[6:9] = dims * rs.uniform(0.9, 1.1, 3)
            row[9:12] = centre + rs.normal(0, 0.05, 3)
            row[12] = yaw + rs.normal(0, 0.08)
            row[13] = rs.uniform(0.8, 1.0)
            row[78:82] = bb
            rows.append(row)
        if rows:
            rows.sort(key=lambda r: r[0])
            tracks.append(np.asarray(rows))
    return {"img_names": img_names, "T_wcs": T_wcs, "P_cws": P_cws, "tracks": tracks, "K": K,
            "img_h": img_h, "img_w": img_w}
