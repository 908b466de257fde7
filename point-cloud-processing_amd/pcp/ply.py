"""Binary little-endian PLY codec compatible with what Open3D's legacy point-cloud writer emits
by default (`double x y z`, `uchar red green blue`) — the inter-stage file format of the
reference (XPC:113-123, DP3:54-58,75-81, CHK:135-142).  Open3D itself is not required.

Colours are quantised on write as round(clamp(c,0,1)*255) and returned on read as uchar/255.0
(float64), which is what the reference's stage 2 and chunking script see."""
import numpy as np

_TYPES = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "<i2", "ushort": "<u2",
          "int16": "<i2", "uint16": "<u2", "int": "<i4", "uint": "<u4", "int32": "<i4",
          "uint32": "<u4", "float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8"}


def write_point_cloud(path, xyz, rgb=None):
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    n = xyz.shape[0]
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    header = ["ply", "format binary_little_endian 1.0", "comment Created by pcp (Open3D layout)",
              f"element vertex {n}", "property double x", "property double y", "property double z"]
    if rgb is not None:
        rgb = np.asarray(rgb, dtype=np.float64).reshape(-1, 3)
        if rgb.shape[0] != n:
            raise ValueError("xyz and rgb row counts differ")
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += ["property uchar red", "property uchar green", "property uchar blue"]
    header.append("end_header")
    rec = np.empty(n, dtype=np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if rgb is not None:
        q = np.rint(np.clip(rgb, 0.0, 1.0) * 255.0).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = q[:, 0], q[:, 1], q[:, 2]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(rec.tobytes())


def read_point_cloud(path):
    """-> (points (N,3) float64, colors (N,3) float64 in [0,1] or None)."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on vertices are not supported")
                props.append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        if fmt in ("binary_little_endian", "binary_big_endian"):   # Open3D's reader (rply) takes both
            order = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(name, _TYPES[t].replace("<", order)) for name, t in props])
            rec = np.frombuffer(fh.read(n * dt.itemsize), dtype=dt, count=n)
            col = lambda k: rec[k]
        elif fmt == "ascii":
            arr = np.loadtxt(fh, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            names = [p[0] for p in props]
            col = lambda k: arr[:, names.index(k)]
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    names = [p[0] for p in props]
    pts = np.stack([np.asarray(col(k), dtype=np.float64) for k in ("x", "y", "z")], axis=1) if n else np.zeros((0, 3))
    colors = None
    if all(k in names for k in ("red", "green", "blue")):
        ctype = dict(props)["red"]
        raw = np.stack([np.asarray(col(k), dtype=np.float64) for k in ("red", "green", "blue")], axis=1) if n else np.zeros((0, 3))
        colors = raw / 255.0 if ctype in ("uchar", "uint8") else raw
    return pts, colors
