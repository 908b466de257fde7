"""Seeded synthetic RGB-D frames of the benchmark's shape: depth is u16
millimetres uniform in [300,1500) converted to f32 metres exactly as the topic extractor does
(extract_topics_one_camera.py:68-72), rgb is uniform u8.  scene_frame() is the second distribution of
SURVEY.md §8d: a tabletop with boxes in the robot-base frame, ray-cast into the stage-1 camera — smooth
surfaces, so the cropped cloud is a 2-D sheet of ~25-40 % of the pixels instead of a random volume fill."""
import numpy as np

# base <- camera pose the stage-1 driver passes (pipeline_stage1_extract.py:13-18)
S1_EXTRINSICS = np.array([[-0.5777, 0.4385, -0.6884, 0.7814],
                          [0.7992, 0.1327, -0.5862, 0.4271],
                          [-0.1657, -0.8889, -0.4271, 0.5267],
                          [0.0, 0.0, 0.0, 1.0]])


def frame_mm(seed, H=480, W=640, Hc=None, Wc=None):
    """The raw form of frame(): depth as the u16 millimetres a depth topic carries (XTP:68-72 divides by 1000)."""
    rng = np.random.default_rng(seed)
    depth_mm = rng.integers(300, 1500, (H, W)).astype(np.uint16)
    rgb = rng.integers(0, 256, (Hc or H, Wc or W, 3), dtype=np.uint8)
    return depth_mm, rgb


def frame(seed, H=480, W=640, Hc=None, Wc=None):
    depth_mm, rgb = frame_mm(seed, H, W, Hc, Wc)
    depth = (depth_mm.astype(np.float32) / 1000.0).astype(np.float32)
    return depth, rgb


def episode_mm(n_frames, seed0=0, H=480, W=640):
    depth = np.empty((n_frames, H, W), np.uint16)
    rgb = np.empty((n_frames, H, W, 3), np.uint8)
    for i in range(n_frames):
        depth[i], rgb[i] = frame_mm(seed0 + i, H, W)
    return depth, rgb


def episode(n_frames, seed0=0, H=480, W=640):
    depth = np.empty((n_frames, H, W), np.float32)
    rgb = np.empty((n_frames, H, W, 3), np.uint8)
    for i in range(n_frames):
        depth[i], rgb[i] = frame(seed0 + i, H, W)
    return depth, rgb


def intrinsics(H=480, W=640, f=605.0):
    return dict(fx=f, fy=f, cx=W / 2, cy=H / 2)


def scene_frame(seed, H=480, W=640, f=605.0, ext=None):
    """Depth (f32 metres from u16 millimetres, 0 = no return) and rgb of a table plane z = 0 carrying 3-6
    axis-aligned boxes, seen from `ext` (base <- camera).  1 mm Gaussian range noise, 2 % dropped pixels."""
    rng = np.random.default_rng(seed)
    E = S1_EXTRINSICS if ext is None else np.asarray(ext, np.float64).reshape(4, 4)
    R, o = E[:3, :3], E[:3, 3]
    v, u = np.mgrid[0:H, 0:W]
    dc = np.stack([(u - W / 2) / f, (v - H / 2) / f, np.ones((H, W))], -1)       # camera-frame rays, z = 1
    dw = dc @ R.T
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(dw[..., 2] < 0, -o[2] / dw[..., 2], np.inf)                 # table plane z = 0
        hit = o + t[..., None] * dw
        on_table = (np.abs(hit[..., 0] - 0.2) < 0.6) & (np.abs(hit[..., 1] - 0.1) < 0.6)
        t = np.where(on_table, t, np.inf)
        col = np.where(on_table[..., None], np.array([150, 120, 90]), 0).astype(np.float64)
        for _ in range(int(rng.integers(3, 7))):
            c = np.array([rng.uniform(0.0, 0.35), rng.uniform(-0.08, 0.3)])
            s = rng.uniform(0.02, 0.06, 2)
            lo = np.array([c[0] - s[0], c[1] - s[1], 0.0])
            hi = np.array([c[0] + s[0], c[1] + s[1], rng.uniform(0.03, 0.15)])
            t1, t2 = (lo - o) / dw, (hi - o) / dw                                # slab method
            tn, tf = np.minimum(t1, t2).max(-1), np.maximum(t1, t2).min(-1)
            ok = (tn <= tf) & (tn > 0) & (tn < t)
            t = np.where(ok, tn, t)
            col = np.where(ok[..., None], rng.integers(40, 256, 3), col)
    mm = t * 1000.0 + rng.normal(0.0, 1.0, (H, W))
    mm = np.where(np.isfinite(mm) & (mm < 60000) & (rng.random((H, W)) > 0.02), mm, 0.0)
    depth = (np.rint(mm).astype(np.uint16).astype(np.float32) / 1000.0).astype(np.float32)
    rgb = np.clip(col + rng.normal(0, 6, (H, W, 3)), 0, 255).astype(np.uint8)
    return depth, rgb


def _scene_job(a):
    return scene_frame(*a)


def scene_episode(n_frames, seed0=0, H=480, W=640, f=605.0, workers=8):
    depth = np.empty((n_frames, H, W), np.float32)
    rgb = np.empty((n_frames, H, W, 3), np.uint8)
    jobs = [(seed0 + i, H, W, f) for i in range(n_frames)]
    if workers > 1 and n_frames >= 16:                       # ray casting is ~0.4 s per frame on one core
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            frames = pool.map(_scene_job, jobs, chunksize=4)
    else:
        frames = [_scene_job(j) for j in jobs]
    for i, (dd, cc) in enumerate(frames):
        depth[i], rgb[i] = dd, cc
    return depth, rgb
