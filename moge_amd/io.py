"""Host-side writers for the caller step after `infer()` (reference: moge/utils/io.py:18-64, which delegates to trimesh -
not installed here, so the byte layout below is this package's own: standard binary little-endian PLY)."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Union

import numpy as np


def masked_point_cloud(points: np.ndarray, mask: np.ndarray, image: Optional[np.ndarray] = None, normal: Optional[np.ndarray] = None):
    """Vertices (N,3) / colors (N,3) float in [0,1] / normals of the valid pixels of one image, in the export convention of
    scripts/infer.py:146-149 (OpenGL: x right, y up, z backward)."""
    m = mask.astype(bool)
    v = points[m].astype(np.float32) * np.array([1, -1, -1], dtype=np.float32)
    c = None if image is None else (image[m].astype(np.float32) / 255 if image.dtype == np.uint8 else image[m].astype(np.float32))
    n = None if normal is None else normal[m].astype(np.float32) * np.array([1, -1, -1], dtype=np.float32)
    return v, c, n


def save_ply(path: Union[str, Path], vertices: np.ndarray, faces: Optional[np.ndarray] = None, vertex_colors: Optional[np.ndarray] = None,
             vertex_normals: Optional[np.ndarray] = None) -> None:
    """Same argument order as the reference's `save_ply(save_path, vertices, faces, vertex_colors, vertex_normals)` (moge/utils/io.py:45-64):
    colors are floats in [0,1] and are stored as uchar RGB, like trimesh does."""
    vertices = np.asarray(vertices, dtype="<f4").reshape(-1, 3)
    n = vertices.shape[0]
    fields, cols = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")], [vertices]
    if vertex_normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
        cols.append(np.asarray(vertex_normals, dtype="<f4").reshape(n, 3))
    if vertex_colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        cols.append(np.clip(np.asarray(vertex_colors, dtype=np.float32).reshape(n, 3) * 255, 0, 255).astype("u1"))
    rec = np.empty(n, dtype=fields)
    k = 0
    for c in cols:
        for j in range(c.shape[1]):
            rec[fields[k][0]] = c[:, j]
            k += 1
    faces = np.zeros((0, 3), dtype="<i4") if faces is None else np.asarray(faces, dtype="<i4").reshape(-1, 3)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property {'uchar' if t == 'u1' else 'float'} {name}" for name, t in fields]
    header += [f"element face {faces.shape[0]}", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        rec.tofile(f)
        if faces.shape[0]:
            frec = np.empty(faces.shape[0], dtype=[("n", "u1"), ("v", "<i4", (3,))])
            frec["n"] = 3
            frec["v"] = faces
            frec.tofile(f)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Image mesh (scripts/infer.py:128-145: utils3d.np.build_mesh_from_map + uv_map; utils3d is an un-vendored dependency, so the algorithm is
# restated from the call site: "parity unpinned", like depth_map_edge)
# ---------------------------------------------------------------------------------------------------------------------------------------
def depth_map_edge(depth: np.ndarray, rtol: float, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """utils3d.np.depth_map_edge as the reference's callers use it: a pixel is an edge when the spread of depth in its 3x3 neighbourhood
    (max - min; positions outside the image do not take part) exceeds rtol x its own depth.  Without `mask` (scripts/infer.py:127) every in-image
    neighbour counts, so a valid pixel next to a +inf (masked-out) depth is an edge; with `mask` (infer_baseline.py:123 passes `mask=mask`)
    masked-out neighbours are excluded from both pools like out-of-image ones, and masked-out pixels are never edges.  Host-side numpy for callers
    that hold no model handle; `MoGeModel.depth_edge_mask` is the same test on the device (it skips NaN neighbours: hand it NaN at masked pixels for
    the `mask=` semantics)."""
    def pool(a):
        p = np.pad(a, 1, mode="constant", constant_values=-np.inf)
        H, W = a.shape
        return np.max(np.stack([p[i:i + H, j:j + W] for i in range(3) for j in range(3)]), axis=0)
    depth = np.asarray(depth)
    with np.errstate(all="ignore"):
        if mask is None:
            return (pool(depth) + pool(-depth)) / depth > rtol
        m = np.asarray(mask).astype(bool)
        hi = pool(np.where(m, depth, -np.inf))
        lo = pool(np.where(m, -depth, -np.inf))
        return m & ((hi + lo) / depth > rtol)


def uv_map(height: int, width: int) -> np.ndarray:
    """Pixel-centre texture coordinates (H, W, 2), u right / v down in [0, 1] (utils3d.np.uv_map as the reference's caller uses it: the
    export step flips v afterwards, scripts/infer.py:149)."""
    u = (np.arange(width, dtype=np.float32) + 0.5) / width
    v = (np.arange(height, dtype=np.float32) + 0.5) / height
    return np.stack(np.meshgrid(u, v, indexing="xy"), axis=-1)


def build_mesh_from_map(*maps: np.ndarray, mask: Optional[np.ndarray] = None, tri: bool = True):
    """Grid mesh over an (H, W) image: one vertex per pixel, one quad per 2x2 pixel block whose four pixels are all inside `mask`, split
    into two triangles when `tri`; vertices no face references are dropped and the faces re-indexed.  Returns (faces, *attributes) with one
    (N, C) attribute array per input map, in the input order - the call shape of `utils3d.np.build_mesh_from_map(points, colors, uvs[,
    normals], mask=..., tri=True)` at scripts/infer.py:129-145."""
    H, W = maps[0].shape[:2]
    m = np.ones((H, W), dtype=bool) if mask is None else mask.astype(bool)
    quad_ok = m[:-1, :-1] & m[:-1, 1:] & m[1:, :-1] & m[1:, 1:]
    idx = np.arange(H * W, dtype=np.int64).reshape(H, W)
    a, b, c, d = idx[:-1, :-1][quad_ok], idx[1:, :-1][quad_ok], idx[1:, 1:][quad_ok], idx[:-1, 1:][quad_ok]      # counter-clockwise in image space
    faces = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], 0) if tri else np.stack([a, b, c, d], -1)
    used = np.zeros(H * W, dtype=bool)
    used[faces.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    faces = remap[faces].astype(np.int32)
    return (faces,) + tuple(np.asarray(x).reshape(H * W, -1)[used] for x in maps)


# ---------------------------------------------------------------------------------------------------------------------------------------
# glTF 2.0 binary (.glb).  The reference hands the mesh to trimesh (moge/utils/io.py:18-42: TextureVisuals + PBRMaterial with
# baseColorTexture = the image, metallicFactor 0.5, roughnessFactor 1.0); trimesh is not installed, so the container is written directly:
# 12-byte header, one JSON chunk, one BIN chunk (positions, normals, uvs, uint32 indices, the texture as PNG).
# ---------------------------------------------------------------------------------------------------------------------------------------
def save_glb(path: Union[str, Path], vertices: np.ndarray, faces: np.ndarray, vertex_uvs: np.ndarray, texture: np.ndarray,
             vertex_normals: Optional[np.ndarray] = None) -> None:
    """Same argument order as the reference's `save_glb(save_path, vertices, faces, vertex_uvs, texture, vertex_normals)`."""
    import io as _io
    import json
    import struct

    from PIL import Image

    vertices = np.ascontiguousarray(vertices, dtype="<f4").reshape(-1, 3)
    faces = np.ascontiguousarray(faces, dtype="<u4").reshape(-1)
    # glTF's texture origin is the top-left corner; the caller passes OpenGL-convention uvs (v up, scripts/infer.py:149): flip v back
    uvs = np.ascontiguousarray(np.asarray(vertex_uvs, dtype="<f4").reshape(-1, 2) * np.array([1, -1], dtype="<f4") + np.array([0, 1], dtype="<f4"))
    png = _io.BytesIO()
    Image.fromarray(np.asarray(texture, dtype=np.uint8)).save(png, format="PNG")
    blobs, views, accessors = [], [], []

    def add(data: bytes, target=None):
        off = sum(len(b) for b in blobs)
        blobs.append(data + b"\x00" * (-len(data) % 4))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if target is not None:
            v["target"] = target
        views.append(v)
        return len(views) - 1

    attrs = {}
    n = vertices.shape[0]
    accessors.append({"bufferView": add(vertices.tobytes(), 34962), "componentType": 5126, "count": n, "type": "VEC3",
                      "min": [float(x) for x in (vertices.min(0) if n else np.zeros(3))], "max": [float(x) for x in (vertices.max(0) if n else np.zeros(3))]})
    attrs["POSITION"] = 0
    if vertex_normals is not None:
        nrm = np.ascontiguousarray(vertex_normals, dtype="<f4").reshape(-1, 3)
        accessors.append({"bufferView": add(nrm.tobytes(), 34962), "componentType": 5126, "count": n, "type": "VEC3"})
        attrs["NORMAL"] = len(accessors) - 1
    accessors.append({"bufferView": add(uvs.tobytes(), 34962), "componentType": 5126, "count": n, "type": "VEC2"})
    attrs["TEXCOORD_0"] = len(accessors) - 1
    accessors.append({"bufferView": add(faces.tobytes(), 34963), "componentType": 5125, "count": int(faces.size), "type": "SCALAR"})
    idx_acc = len(accessors) - 1
    img_view = add(png.getvalue())
    gltf = {
        "asset": {"version": "2.0", "generator": "moge_amd.io.save_glb"},
        "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": attrs, "indices": idx_acc, "material": 0, "mode": 4}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.5, "roughnessFactor": 1.0}, "doubleSided": True}],
        "textures": [{"source": 0, "sampler": 0}], "samplers": [{"magFilter": 9729, "minFilter": 9987, "wrapS": 10497, "wrapT": 10497}],
        "images": [{"bufferView": img_view, "mimeType": "image/png"}],
        "buffers": [{"byteLength": sum(len(b) for b in blobs)}], "bufferViews": views, "accessors": accessors,
    }
    js = json.dumps(gltf, separators=(",", ":")).encode("utf-8")
    js += b" " * (-len(js) % 4)
    binc = b"".join(blobs)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(binc)))
        f.write(struct.pack("<I4s", len(js), b"JSON")); f.write(js)
        f.write(struct.pack("<I4s", len(binc), b"BIN\x00")); f.write(binc)


# ---------------------------------------------------------------------------------------------------------------------------------------
# OpenEXR, float32 scan lines, no compression - what `cv2.imwrite(path, array, [IMWRITE_EXR_TYPE, IMWRITE_EXR_TYPE_FLOAT])` stores for
# depth.exr (one channel) and points.exr (scripts/infer.py:113,115; cv2 is not installed).  Channel naming follows cv2: a 2-D array becomes
# channel "Y"; an (H, W, 3) array in RGB order becomes R, G, B (the reference converts points to BGR for cv2, which then names the last cv2
# channel R: the file's R, G, B are x, y, z).
# ---------------------------------------------------------------------------------------------------------------------------------------
def save_exr(path: Union[str, Path], array: np.ndarray) -> None:
    import struct
    a = np.asarray(array, dtype="<f4")
    if a.ndim == 2:
        chans = {"Y": a}
    elif a.ndim == 3 and a.shape[2] == 3:
        chans = {"R": a[..., 0], "G": a[..., 1], "B": a[..., 2]}
    else:
        raise ValueError("save_exr expects (H, W) or (H, W, 3)")
    H, W = a.shape[:2]
    names = sorted(chans)                                   # the channel list and the pixel data are in alphabetical channel order

    def attr(name, typ, data):
        return name.encode() + b"\x00" + typ.encode() + b"\x00" + struct.pack("<i", len(data)) + data

    chlist = b"".join(n.encode() + b"\x00" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in names) + b"\x00"      # pixel type 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    header = (attr("channels", "chlist", chlist) + attr("compression", "compression", b"\x00") + attr("dataWindow", "box2i", box)
              + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\x00") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\x00")
    line_bytes = len(names) * W * 4
    start = 8 + len(header) + 8 * H
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 20000630, 2))            # magic, version 2, single-part scan-line flags
        f.write(header)
        f.write(np.asarray([start + y * (8 + line_bytes) for y in range(H)], dtype="<u8").tobytes())
        planes = np.stack([chans[n] for n in names], axis=1)            # (H, C, W): per scan line, channel after channel
        for y in range(H):
            f.write(struct.pack("<ii", y, line_bytes))
            f.write(planes[y].tobytes())


def read_exr(path: Union[str, Path]) -> np.ndarray:
    """Reader for the files `save_exr` writes (uncompressed float32 scan lines): -> (H, W) for a single channel, else (H, W, 3) in R, G, B order."""
    import struct
    with open(path, "rb") as f:
        buf = f.read()
    assert struct.unpack_from("<I", buf, 0)[0] == 20000630
    p, names, box = 8, [], None
    while buf[p] != 0:
        e = buf.index(b"\x00", p); name = buf[p:e].decode(); p = e + 1
        e = buf.index(b"\x00", p); p = e + 1
        size = struct.unpack_from("<i", buf, p)[0]; p += 4
        data = buf[p:p + size]; p += size
        if name == "channels":
            q = 0
            while data[q] != 0:
                e = data.index(b"\x00", q); names.append(data[q:e].decode()); q = e + 1 + 16
        elif name == "dataWindow":
            box = struct.unpack("<iiii", data)
        elif name == "compression":
            assert data == b"\x00", "only uncompressed files"
    p += 1
    W, H = box[2] - box[0] + 1, box[3] - box[1] + 1
    offs = np.frombuffer(buf, dtype="<u8", count=H, offset=p)
    out = np.empty((H, len(names), W), dtype=np.float32)
    for y in range(H):
        o = int(offs[y]) + 8
        out[y] = np.frombuffer(buf, dtype="<f4", count=len(names) * W, offset=o).reshape(len(names), W)
    if len(names) == 1:
        return out[:, 0]
    return np.stack([out[:, names.index(c)] for c in ("R", "G", "B")], axis=-1)


def colorize_normal(normal: np.ndarray, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """moge/utils/vis.py:53-58."""
    if mask is not None:
        normal = np.where(mask[..., None], normal, 0)
    return ((normal * [0.5, -0.5, -0.5] + 0.5).clip(0, 1) * 255).astype(np.uint8)


# matplotlib's "Spectral" colour map (11 ColorBrewer anchors, linearly interpolated - what matplotlib.colormaps['Spectral'] evaluates)
_SPECTRAL = np.array([[158, 1, 66], [213, 62, 79], [244, 109, 67], [253, 174, 97], [254, 224, 139], [255, 255, 191], [230, 245, 152],
                      [171, 221, 164], [102, 194, 165], [50, 136, 189], [94, 79, 162]], dtype=np.float64) / 255.0


def colorize_depth(depth: np.ndarray, mask: Optional[np.ndarray] = None, normalize: bool = True) -> np.ndarray:
    """moge/utils/vis.py:7-18 (Spectral colour map over normalised disparity; matplotlib is not installed, the map is tabulated above)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        # exactly the reference's filter: an infinite depth (infer()'s masked pixels with apply_mask=True) PASSES `depth > 0`, becomes disparity
        # 0, takes part in the quantile normalisation and is painted the colour map's far end - not black
        d = np.where((depth > 0) & (True if mask is None else mask), depth, np.nan)
        disp = 1 / d
        if normalize and np.isfinite(disp).any():
            lo, hi = np.nanquantile(disp, 0.001), np.nanquantile(disp, 0.99)
            disp = (disp - lo) / (hi - lo)
        t = np.clip(1.0 - disp, 0, 1) * (len(_SPECTRAL) - 1)
    i0 = np.clip(np.floor(np.nan_to_num(t)).astype(int), 0, len(_SPECTRAL) - 2)
    w = (np.nan_to_num(t) - i0)[..., None]
    col = _SPECTRAL[i0] * (1 - w) + _SPECTRAL[i0 + 1] * w
    col = np.where(np.isnan(t)[..., None], 0.0, col)
    return np.ascontiguousarray((col.clip(0, 1) * 255).astype(np.uint8))
