"""`python -m moge_amd.scripts.infer_panorama` - the reference's `moge infer_panorama` (moge/scripts/infer_panorama.py:14-160) on the MI355X path.

Same flags where the step exists here (`--input/-i`, `--output/-o`, `--pretrained`, `--device`, `--resize`, `--resolution_level`, `--threshold`,
`--batch_size`, `--splitted`, `--maps`, `--glb`, `--ply`), plus `--version v1|v2` (the reference script is MoGe-1 only) and `--fp16`.  The
equirectangular image is split into 12 views, the views go through `MoGeModel.infer(views, fov_x=90, apply_mask=False)` in batches on the
GPU, the distance maps are merged on the host (moge_amd/panorama.py).  Differences, all forced by what this image ships: decode / resize use
PIL (BOX filter for `--resize`), EXR / GLB / PLY are written by moge_amd.io, `--show` is accepted and warns (no viewer here), and the mesh mask is
`mask & ~depth_map_edge(distance, rtol=threshold)` - the reference additionally requires a normal-map edge (utils3d.np.normal_map_edge, not
restated here), so this removes a superset of the reference's edge pixels.  `--resolution_level` is accepted but not passed to `infer()`, as in the
reference (infer_panorama.py:101).  File conventions: by default the outputs are the physically consistent ones of `moge infer` (points.exr with
R, G, B = x, y, z; the GLB's uvs in the v-up convention `moge_amd.io.save_glb` expects, so the texture is upright).  The reference's panorama script
differs from its own `scripts/infer.py` in two places - :132 writes `points` through cv2 without the RGB -> BGR conversion of `infer.py:114` (file channels
R, G, B = z, y, x) and :147 passes the mesh builder's uvs unflipped (`infer.py:148` flips v) - and whether the second one shows as a mirrored texture depends on
the un-vendored utils3d / trimesh conventions; `--reference_compat` reproduces both byte conventions for callers that parse the reference's files."""
from __future__ import annotations

import itertools
from pathlib import Path

import click
import numpy as np


@click.command(help="Inference script for panorama images (MI355X)")
@click.option("--input", "-i", "input_path", type=click.Path(exists=True), required=True, help="Input equirectangular image or folder.")
@click.option("--output", "-o", "output_path", type=click.Path(), default="./output", help='Output folder, default "./output".')
@click.option("--pretrained", "pretrained_model_name_or_path", type=str, default="Ruicheng/moge-vitl",
              help='Checkpoint path or Hugging Face repo id. Defaults to "Ruicheng/moge-vitl" as in the reference script (needs network access).')
@click.option("--version", "model_version", type=click.Choice(["v1", "v2"]), default="v1", help="Model class (the reference script uses v1).")
@click.option("--device", "device_name", type=str, default="cuda", help='Device, default "cuda".')
@click.option("--fp16", "use_fp16", is_flag=True, help="fp16 inference (model.half()).")
@click.option("--resize", "resize_to", type=int, default=None, help="Resize the long side to this size first.")
@click.option("--resolution_level", type=int, default=9, help="0-9; accepted and unused, as in the reference script (its infer() call does not pass it).")
@click.option("--threshold", type=float, default=0.03, help="Relative depth-edge threshold of the mesh mask, default 0.03.")
@click.option("--batch_size", type=int, default=4, help="Views per infer() call, default 4.")
@click.option("--splitted", "save_splitted", is_flag=True, help="Also save the 12 views and their distance visualisations.")
@click.option("--maps", "save_maps_", is_flag=True, help="Save image, depth (EXR + visualisation), points (EXR) and mask.")
@click.option("--glb", "save_glb_", is_flag=True, help="Save a textured mesh (.glb).")
@click.option("--ply", "save_ply_", is_flag=True, help="Save a coloured mesh (.ply).")
@click.option("--show", "show", is_flag=True, help="Accepted for compatibility: the reference opens a trimesh viewer here, which this image does not ship.")
@click.option("--reference_compat", "reference_compat", is_flag=True,
              help="Write points.exr (R, G, B = z, y, x) and the GLB uvs (unflipped) exactly as the reference's panorama script does; default: the conventions of `moge infer`.")
def main(input_path, output_path, pretrained_model_name_or_path, model_version, device_name, use_fp16, resize_to, resolution_level, threshold,
         batch_size, save_splitted, save_maps_, save_glb_, save_ply_, show, reference_compat=False):
    import torch
    from PIL import Image

    from moge_amd.io import build_mesh_from_map, colorize_depth, save_exr, save_glb, save_ply, uv_map
    from moge_amd.model import import_model_class_by_version
    from moge_amd.panorama import infer_panorama

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
    if not (save_maps_ or save_glb_ or save_ply_):
        save_maps_ = save_glb_ = save_ply_ = True
    model = import_model_class_by_version(model_version).from_pretrained(pretrained_model_name_or_path).to(torch.device(device_name)).eval()
    if use_fp16:
        model.half()

    for p in image_paths:
        im = Image.open(p).convert("RGB")
        if resize_to is not None:
            w, h = im.size
            h2, w2 = min(resize_to, int(resize_to * h / w)), min(resize_to, int(resize_to * w / h))
            im = im.resize((w2, h2), Image.BOX)
        image = np.asarray(im, dtype=np.uint8)
        H, W = image.shape[:2]
        # (`--resolution_level` is accepted and NOT forwarded: the reference calls `model.infer(image_tensor, fov_x=fov_x, apply_mask=False)` with the
        #  model's default, moge/scripts/infer_panorama.py:101)
        out = infer_panorama(model, image, resolution=512, batch_size=batch_size, merge_size=(1920, 960))
        depth, mask, points = out["distance"], out["mask"], out["points"]
        save_path = Path(output_path, p.relative_to(root).parent, p.stem)
        save_path.mkdir(exist_ok=True, parents=True)
        if save_splitted:
            sp = save_path / "splitted"
            sp.mkdir(exist_ok=True)
            for i, (v, d, m) in enumerate(zip(out["views"], out["view_distance"], out["view_mask"])):
                Image.fromarray(v).save(sp / f"{i:02d}.jpg")
                Image.fromarray(colorize_depth(d, m)).save(sp / f"{i:02d}_distance_vis.png")
        if save_maps_:
            Image.fromarray(image).save(save_path / "image.jpg")
            Image.fromarray(colorize_depth(depth, mask=mask)).save(save_path / "depth_vis.png")
            save_exr(save_path / "depth.exr", depth)
            # R, G, B = x, y, z like `moge infer`; --reference_compat: the reference's PANORAMA script hands `points` to cv2.imwrite as they are
            # (infer_panorama.py:132), cv2 takes the last axis as B, G, R, so its file has R = z, G = y, B = x
            save_exr(save_path / "points.exr", points[..., ::-1] if reference_compat else points)
            Image.fromarray((mask * 255).astype(np.uint8)).save(save_path / "mask.png")
        if save_glb_ or save_ply_:
            cleaned = model.depth_edge_mask(torch.from_numpy(depth)[None], torch.from_numpy(mask)[None], rtol=threshold).cpu().numpy()[0]
            faces, vertices, vertex_colors, vertex_uvs = build_mesh_from_map(points, image.astype(np.float32) / 255, uv_map(H, W), mask=cleaned, tri=True)
            if save_glb_:
                # save_glb takes OpenGL (v-up) uvs and flips them back to glTF's top-left origin: hand it v-up uvs so the texture is upright
                # (--reference_compat: the builder's v-down uvs as infer_panorama.py:147 passes them)
                save_glb(save_path / "mesh.glb", vertices, faces, vertex_uvs if reference_compat else vertex_uvs * [1, -1] + [0, 1], image)
            if save_ply_:
                save_ply(save_path / "mesh.ply", vertices, faces, vertex_colors)


if __name__ == "__main__":
    main()
