"""`python -m moge_amd.scripts.infer` - the reference's `moge infer` caller loop (moge/scripts/infer.py:18-156) on the MI355X path.

Same flags (`--input/-i`, `--output/-o`, `--pretrained` with the reference's per-version default, `--version {v1,v2}`, `--fov_x`, `--resize`,
`--resolution_level`, `--num_tokens`, `--threshold`, `--maps`, `--glb`, `--ply`, `--fp16`, `--device`; `--show` is accepted and warns: no viewer here);
images of equal size are batched (`--batch`) and run through
`moge_amd.pipeline.InferPipeline` (uint8 upload, transfers overlapped with compute).  Differences, all forced by what this image ships:
decode / resize use PIL instead of cv2 (BOX filter for `--resize`, the closest PIL has to INTER_AREA); `depth.exr` / `points.exr` are
written by moge_amd.io.save_exr (uncompressed float32 OpenEXR, same channels as cv2 writes); `mesh.glb` by moge_amd.io.save_glb (glTF 2.0
binary written directly - trimesh is not installed; same material parameters as moge/utils/io.py:18-42).  Mesh and point cloud are built from `mask & ~depth_map_edge(depth, rtol=threshold)` exactly as scripts/infer.py:127-149 does.
"""
from __future__ import annotations

import itertools
import json
import math
from pathlib import Path

import click
import numpy as np


@click.command(help="Inference script (MI355X)")
@click.option("--input", "-i", "input_path", type=click.Path(exists=True), required=True, help="Input image or folder.")
@click.option("--fov_x", "fov_x_", type=float, default=None, help="Horizontal FoV in degrees; recovered from the point map if unset.")
@click.option("--output", "-o", "output_path", default="./output", type=click.Path(), help='Output folder, default "./output".')
@click.option("--pretrained", "pretrained_model_name_or_path", type=str, default=None,
              help='Checkpoint path or Hugging Face repo id. Defaults to "Ruicheng/moge-vitl" (v1) / "Ruicheng/moge-2-vitl-normal" (v2), which needs network access.')
@click.option("--version", "model_version", type=click.Choice(["v1", "v2"]), default="v2", help='Model version. Defaults to "v2".')
@click.option("--device", "device_name", type=str, default="cuda", help='Device, default "cuda".')
@click.option("--fp16", "use_fp16", is_flag=True, help="fp16 inference.")
@click.option("--resize", "resize_to", type=int, default=None, help="Resize the long side to this size before inference.")
@click.option("--resolution_level", type=int, default=9, help="0-9; ignored when --num_tokens is given.")
@click.option("--num_tokens", type=int, default=None, help="Number of ViT tokens, [1200, 2500] suggested.")
@click.option("--threshold", type=float, default=0.04, help="Relative depth-edge threshold for the point cloud, default 0.04.")
@click.option("--maps", "save_maps_", is_flag=True, help="Save depth / points / mask / normal maps and fov.json.")
@click.option("--glb", "save_glb_", is_flag=True, help="Save a textured mesh (.glb).")
@click.option("--ply", "save_ply_", is_flag=True, help="Save a coloured point cloud (.ply).")
@click.option("--show", "show", is_flag=True, help="Accepted for compatibility: the reference opens a trimesh viewer here, which this image does not ship.")
@click.option("--batch", "batch", type=int, default=8, help="Images of equal size per infer() call.")
def main(input_path, fov_x_, output_path, pretrained_model_name_or_path, model_version, device_name, use_fp16, resize_to, resolution_level, num_tokens,
         threshold, save_maps_, save_glb_, save_ply_, show, batch):
    import torch
    from PIL import Image

    from moge_amd.io import build_mesh_from_map, colorize_depth, colorize_normal, save_exr, save_glb, save_ply, uv_map
    from moge_amd.model import import_model_class_by_version
    from moge_amd.pipeline import InferPipeline

    suffices = ["jpg", "png", "jpeg", "JPG", "PNG", "JPEG"]
    if Path(input_path).is_dir():
        image_paths = sorted(itertools.chain(*(Path(input_path).rglob(f"*.{s}") for s in suffices)))
        root = Path(input_path)
    else:
        image_paths, root = [Path(input_path)], Path(input_path).parent
    if len(image_paths) == 0:
        raise FileNotFoundError(f"No image files found in {input_path}")
    if show:
        import warnings
        warnings.warn("--show: no viewer in this environment (trimesh is not installed); the requested files are still written")
    if pretrained_model_name_or_path is None:                         # scripts/infer.py:76-81
        pretrained_model_name_or_path = {"v1": "Ruicheng/moge-vitl", "v2": "Ruicheng/moge-2-vitl-normal"}[model_version]
    model = import_model_class_by_version(model_version).from_pretrained(pretrained_model_name_or_path).to(torch.device(device_name)).eval()
    if use_fp16:
        model.half()
    if not (save_maps_ or save_glb_ or save_ply_):
        save_maps_ = save_glb_ = save_ply_ = True

    def load(path):
        im = Image.open(path).convert("RGB")
        if resize_to is not None:
            w, h = im.size
            h2, w2 = min(resize_to, int(resize_to * h / w)), min(resize_to, int(resize_to * w / h))
            im = im.resize((w2, h2), Image.BOX)
        return np.asarray(im, dtype=np.uint8)

    by_shape = {}
    for p in image_paths:
        with Image.open(p) as im:
            w, h = im.size
        if resize_to is not None:
            h, w = min(resize_to, int(resize_to * h / w)), min(resize_to, int(resize_to * w / h))
        by_shape.setdefault((h, w), []).append(p)

    for (h, w), paths in by_shape.items():
        B = min(batch, len(paths))
        pipe = InferPipeline(model, B, h, w, fov_x=fov_x_, resolution_level=resolution_level, num_tokens=num_tokens, use_fp16=use_fp16)
        chunks = [paths[i:i + B] for i in range(0, len(paths), B)]
        loaded = []

        def gen():
            for ch in chunks:
                imgs = np.stack([load(p) for p in ch])
                loaded.append(imgs)
                yield imgs

        for ch, out in zip(chunks, pipe.run(gen())):
            imgs = loaded.pop(0)
            cleaned = None
            if save_ply_ or save_glb_:
                cleaned = model.depth_edge_mask(torch.from_numpy(out["depth"]), torch.from_numpy(out["mask"]) if "mask" in out else None,
                                                rtol=threshold).cpu().numpy()
            for j, p in enumerate(ch):
                save_path = Path(output_path, p.relative_to(root).parent, p.stem)
                save_path.mkdir(exist_ok=True, parents=True)
                if save_maps_:
                    Image.fromarray(imgs[j]).save(save_path / "image.jpg")
                    Image.fromarray(colorize_depth(out["depth"][j])).save(save_path / "depth_vis.png")
                    save_exr(save_path / "depth.exr", out["depth"][j])
                    save_exr(save_path / "points.exr", out["points"][j])
                    if "mask" in out:
                        Image.fromarray((out["mask"][j] * 255).astype(np.uint8)).save(save_path / "mask.png")
                    if "normal" in out:
                        Image.fromarray(colorize_normal(out["normal"][j])).save(save_path / "normal.png")
                    K = out["intrinsics"][j]
                    with open(save_path / "fov.json", "w") as f:           # normalised intrinsics: fov = 2 atan(0.5 / f)
                        json.dump({"fov_x": round(math.degrees(2 * math.atan(0.5 / float(K[0, 0]))), 2),
                                   "fov_y": round(math.degrees(2 * math.atan(0.5 / float(K[1, 1]))), 2)}, f)
                if save_glb_ or save_ply_:
                    maps = [out["points"][j], imgs[j].astype(np.float32) / 255, uv_map(h, w)] + ([out["normal"][j]] if "normal" in out else [])
                    faces, vertices, vertex_colors, vertex_uvs, *rest = build_mesh_from_map(*maps, mask=cleaned[j], tri=True)
                    # OpenGL conventions for the export (scripts/infer.py:146-151): x right, y up, z backward; (0, 0) = left-bottom of the texture
                    vertices, vertex_uvs = vertices * [1, -1, -1], vertex_uvs * [1, -1] + [0, 1]
                    vertex_normals = rest[0] * [1, -1, -1] if rest else None
                    if save_glb_:
                        save_glb(save_path / "mesh.glb", vertices, faces, vertex_uvs, imgs[j], vertex_normals)
                    if save_ply_:
                        save_ply(save_path / "pointcloud.ply", vertices, np.zeros((0, 3), dtype=np.int32), vertex_colors, vertex_normals)


if __name__ == "__main__":
    main()
