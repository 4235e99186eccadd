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
