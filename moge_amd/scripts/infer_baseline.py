"""`python -m moge_amd.scripts.infer_baseline` - the reference's `moge infer_baseline` (moge/scripts/infer_baseline.py:16-137): run a wrapped
method (a `Baseline` plugin file such as baselines/moge_mi355x.py) over images and write what it returns.

Same flags (`--baseline`, `--input/-i`, `--output/-o`, `--size`, `--skip`, `--maps`, `--ply`, `--glb`, `--threshold`); every other argument is
handed to the plugin's own click command `Baseline.load` (infer_baseline.py:16, 43).  The plugin is loaded BY PATH (tools.py:285) and driven
through the `MGEBaselineInterface` surface only (`device`, `infer(image)`), so any baseline file of the reference's harness works here too.
Outputs per image, as in the reference (:77-112): image.jpg, mask.png, `<points key>.exr`, `<depth key>.exr` + `_vis.png`, fov.json (fov_x, fov_y
in degrees + the intrinsics), mesh.ply / mesh.glb.
Differences, forced by what this image ships: decode / resize use PIL (BOX filter for `--size`), EXR / GLB / PLY are written by moge_amd.io, and
the mesh mask is `mask & ~depth_map_edge(z, rtol=threshold)` where the reference also requires a normal-map edge (utils3d.np.normal_map_edge,
un-vendored: not restated) - a superset of the reference's removed pixels.  One reference quirk is NOT mirrored: its mesh export reads the
`depth` variable left over from the `--maps` loop (:123), so `--ply` without `--maps` raises NameError there; here the depth of the mesh mask is
the point map's z.  `depth_affine_invariant` / `disparity_affine_invariant` get the plain depth colour map (their affine-aware colourings live in
moge/utils/vis.py, which no MoGe plugin output reaches)."""
from __future__ import annotations

import importlib.util
import itertools
import json
import warnings
from pathlib import Path

import click
import numpy as np

POINT_KEYS = ["points_metric", "points_scale_invariant", "points_affine_invariant"]
DEPTH_KEYS = ["depth_metric", "depth_scale_invariant", "depth_affine_invariant", "disparity_affine_invariant"]


def import_file_as_module(file_path, module_name: str):
    """moge/utils/tools.py:285-289"""
    spec = importlib.util.spec_from_file_location(module_name, file_path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


@click.command(context_settings={"allow_extra_args": True, "ignore_unknown_options": True}, help="Inference script for wrapped baseline methods (MI355X)")
@click.option("--baseline", "baseline_code_path", required=True, type=click.Path(), help="Path to the baseline model python code.")
@click.option("--input", "-i", "input_path", type=str, required=True, help="Input image or folder")
@click.option("--output", "-o", "output_path", type=str, default="./output", help="Output folder")
@click.option("--size", "image_size", type=int, default=None, help="Resize input image")
@click.option("--skip", is_flag=True, help="Skip existing output")
@click.option("--maps", "save_maps_", is_flag=True, help="Save output point / depth maps")
@click.option("--ply", "save_ply_", is_flag=True, help="Save mesh in PLY format")
@click.option("--glb", "save_glb_", is_flag=True, help="Save mesh in GLB format")
@click.option("--threshold", type=float, default=0.03, help="Depth edge detection threshold for saving mesh")
@click.pass_context
def main(ctx, baseline_code_path, input_path, output_path, image_size, skip, save_maps_, save_ply_, save_glb_, threshold):
    import torch
    from PIL import Image

    from moge_amd.io import build_mesh_from_map, colorize_depth, depth_map_edge, save_exr, save_glb, save_ply, uv_map

    module = import_file_as_module(baseline_code_path, Path(baseline_code_path).stem)
    baseline = getattr(module, "Baseline").load.main(ctx.args, standalone_mode=False)

    suffices = ["jpg", "png", "jpeg", "JPG", "PNG", "JPEG"]
    if Path(input_path).is_dir():
        image_paths = sorted(itertools.chain(*(Path(input_path).rglob(f"*.{s}") for s in suffices)))
        root = Path(input_path)
    else:
        image_paths, root = [Path(input_path)], Path(input_path).parent
    if not (save_maps_ or save_glb_ or save_ply_):
        warnings.warn('No output format specified. Defaults to saving maps only. Please use "--maps", "--glb", or "--ply" to specify the output.')
        save_maps_ = True

    for image_path in image_paths:
        save_path = Path(output_path, image_path.relative_to(root).parent, image_path.stem)
        if skip and save_path.exists():
            continue
        im = Image.open(image_path).convert("RGB")
        width, height = im.size
        if image_size is not None and max(height, width) > image_size:
            height, width = min(image_size, int(image_size * height / width)), min(image_size, int(image_size * width / height))
            im = im.resize((width, height), Image.BOX)
        image_np = np.asarray(im, dtype=np.uint8)
        image = torch.from_numpy(image_np.astype(np.float32) / 255.0).permute(2, 0, 1).to(baseline.device)
        with torch.inference_mode():
            output = baseline.infer(image)
        output = {k: v.cpu().numpy() if isinstance(v, torch.Tensor) else v for k, v in output.items()}
        save_path.mkdir(parents=True, exist_ok=True)

        if save_maps_:
            Image.fromarray(image_np).save(save_path / "image.jpg")
            if "mask" in output:
                Image.fromarray((output["mask"] * 255).astype(np.uint8)).save(save_path / "mask.png")
            for k in POINT_KEYS:
                if k in output:
                    save_exr(save_path / f"{k}.exr", output[k])
            for k in DEPTH_KEYS:
                if k in output:
                    save_exr(save_path / f"{k}.exr", output[k])
                    Image.fromarray(colorize_depth(output[k])).save(save_path / f"{k}_vis.png")
            if "intrinsics" in output:
                K = output["intrinsics"]
                with open(save_path / "fov.json", "w") as f:           # normalised intrinsics: fov = 2 atan(0.5 / f)   (geometry_numpy.py intrinsics_to_fov_numpy)
                    json.dump({"fov_x": float(np.rad2deg(2 * np.arctan(0.5 / K[0, 0]))), "fov_y": float(np.rad2deg(2 * np.arctan(0.5 / K[1, 1]))),
                               "intrinsics": K.tolist()}, f, indent=4)

        if save_ply_ or save_glb_:
            assert any(k in output for k in POINT_KEYS), "No point map found in output"
            points = next(output[k] for k in POINT_KEYS if k in output)
            # (no mask from the plugin: all-ones, like infer_baseline.py:119)
            mask = output["mask"].astype(bool) if "mask" in output else np.ones(points.shape[:-1], dtype=bool)
            # depth_map_edge(depth, rtol, mask=mask) (infer_baseline.py:123): masked-out neighbours take no part in the 3x3 max / min, so the mesh is NOT
            # eroded along mask borders.  The device kernel skips NaN neighbours (fmaxf / fminf), which gives the same semantics with NaN at masked pixels.
            on_device = getattr(getattr(baseline, "model", None), "depth_edge_mask", None)          # a moge_amd model behind the plugin: the device kernel
            if on_device is not None:
                z = np.where(mask, points[..., 2], np.nan).astype(np.float32)
                clean = on_device(torch.from_numpy(z), torch.from_numpy(mask), rtol=threshold).cpu().numpy()
            else:
                clean = mask & ~depth_map_edge(points[..., 2].astype(np.float32), threshold, mask=mask)
            faces, vertices, vertex_colors, vertex_uvs = build_mesh_from_map(np.where(mask[..., None], points, 0).astype(np.float32), image_np.astype(np.float32) / 255,
                                                                             uv_map(height, width), mask=clean, tri=True)
            # OpenGL conventions for the export (infer_baseline.py:127-130): x right, y up, z backward; (0, 0) = left-bottom of the texture
            vertices, vertex_uvs = vertices * [1, -1, -1], vertex_uvs * [1, -1] + [0, 1]
            if save_glb_:
                save_glb(save_path / "mesh.glb", vertices, faces, vertex_uvs, image_np)
            if save_ply_:
                save_ply(save_path / "mesh.ply", vertices, faces, vertex_colors)


if __name__ == "__main__":
    main()
