"""CPU: the C-ABI library loads without a GPU and exports exactly what include/moge_hip.h declares; host-side
shape logic of the Python mirror matches the reference's rules.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_match_header():
    from moge_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "moge_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(moge_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(L.EXPORTS)
    for name in declared:
        assert hasattr(L.lib, name), f"libmoge_hip.so does not export {name}"
    assert L.lib.moge_abi_version() == L.ABI_VERSION == 5


def test_config_struct_layout():
    from moge_amd import _lib as L
    assert ctypes.sizeof(L.MogeConfig) == 4 * (4 + 8 + 5 + 5 + 5 + 3 + 4 + 4 + 4 + 4)      # ABI v4: + activations, hidden multipliers
    assert ctypes.sizeof(L.Outputs) == 9 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(L.Profile) == 8 * 8 * 4


def test_model_mirror_host_logic():
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle as O
    M = import_model_class_by_version("v2")
    m = M(**O.named_configs()["moge-2-vitl-normal"])
    assert m._grid(518, 518, 3600) == (60, 60) and m._grid(518, 1036, 3600) == (42, 85) and m._grid(1036, 518, 3600) == (85, 42)
    assert m._cfg.embed_dim == 1024 and m._cfg.depth == 24 and list(m._cfg.taps)[:4] == [5, 11, 17, 23]
    assert m.num_tokens_range == [1200, 3600] and m.dtype.is_floating_point and m.device.type == "cpu"
    M1 = import_model_class_by_version("v1")                       # MoGe-1 mirror (moge/model/v1.py): same loader, its own config keywords
    from oracle import moge_oracle_v1 as O1
    m1 = M1(**O1.named_configs()["moge-vitl"])
    assert m1._cfg.n_taps == 4 and list(m1._cfg.taps)[:4] == [20, 21, 22, 23] and list(m1._cfg.dim_upsample)[:3] == [256, 128, 128]
    assert m1._resized(518, 518, 2500) == (700, 700) and m1.num_tokens_range == [1200, 2500] and m1.mask_threshold == 0.5
    with pytest.raises(AssertionError):
        import_model_class_by_version("v3")
    with pytest.raises(ValueError):
        M(**{**O.named_configs()["tiny-vits-normal"], "remap_output": "bogus"})
    with pytest.raises(RuntimeError):
        m.to("cpu")            # no CPU fallback, by design
    # the rest of the class surface (v2.py:109-136, v1.py:244-267): training switches are accepted / refused, the remap is the reference's
    import torch
    assert m.enable_pytorch_native_sdpa() is None
    with pytest.warns(UserWarning):
        m.enable_gradient_checkpointing()
    with pytest.raises(NotImplementedError):
        m1.init_weights()
    p = torch.tensor([[0.3, -0.2, 0.5]])
    assert torch.allclose(m._remap_points(p), torch.tensor([[0.3 * 0.5 ** 0 * torch.e ** 0.5, -0.2 * torch.e ** 0.5, torch.e ** 0.5]]))       # 'exp'
    for mode, want in (("linear", p), ("sinh", torch.sinh(p)), ("sinh_exp", torch.cat([torch.sinh(p[:, :2]), torch.exp(p[:, 2:])], -1))):
        mm = M(**{**O.named_configs()["tiny-vits-normal"], "remap_output": mode})
        assert torch.allclose(mm._remap_points(p), want) and torch.allclose(O.remap_points(p, mode), want)


def test_mirror_maps_every_convstack_option_it_supports():
    """modules.py:139-181, 31-60, 199-203: the x2 up-samplers, residual-block norms (ABI v3), activations, InstanceNorm2d and the hidden-width
    multiplier (ABI v4) of ConvStack reach the C ABI as codes; the x0.5 resamplers - which make the reference's own forward a shape error
    (v2.py:154-160 with modules.py:247-249) - and unknown names are refused when the model is constructed."""
    import copy
    from moge_amd import _lib as L
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle as O
    M = import_model_class_by_version("v2")
    cfg = O.named_configs()["tiny-generic-stack"]
    m = M(**cfg)
    assert list(m._cfg.neck_resamplers) == [L.RESAMPLER[r] for r in cfg["neck"]["resamplers"]] == [3, 2, 1, 0]
    assert list(m._cfg.head_resamplers) == [2, 0, 3, 1]
    assert (m._cfg.neck_in_norm, m._cfg.neck_hidden_norm, m._cfg.head_in_norm, m._cfg.head_hidden_norm) == (1, 2, 0, 1)
    assert (m._cfg.neck_activation, m._cfg.head_activation, m._cfg.neck_hidden_mult, m._cfg.head_hidden_mult) == (0, 0, 1, 1)
    rel = M(**O.named_configs()["moge-2-vitl-normal"])._cfg
    assert list(rel.neck_resamplers) == list(rel.head_resamplers) == [0, 0, 0, 1] and rel.neck_in_norm == rel.head_hidden_norm == 0
    assert rel.neck_activation == rel.head_activation == 0 and rel.neck_hidden_mult == rel.head_hidden_mult == 1
    opt = M(**O.named_configs()["tiny-block-options"])._cfg
    assert (opt.neck_in_norm, opt.neck_hidden_norm, opt.head_in_norm, opt.head_hidden_norm) == (3, 2, 0, 3)
    assert (opt.neck_activation, opt.head_activation, opt.neck_hidden_mult, opt.head_hidden_mult) == (L.ACTIVATION["silu"], L.ACTIVATION["elu"], 2, 1)
    optb = M(**O.named_configs()["tiny-block-options-b"])._cfg
    assert (optb.neck_activation, optb.head_activation, optb.neck_hidden_mult, optb.head_hidden_mult) == (0, L.ACTIVATION["leaky_relu"], 2, 4)
    for path, value, exc in ((("neck", "resamplers"), ["conv_transpose", "avg_pool", "conv_transpose", "bilinear"], NotImplementedError),
                             (("neck", "resamplers"), ["pixel_unshuffle"] * 4, NotImplementedError),
                             (("points_head", "res_block_in_norm"), "batch_norm", NotImplementedError),
                             (("neck", "activation"), "gelu", ValueError),                               # modules.py:41 raises ValueError too
                             (("mask_head", "dim_times_res_block_hidden"), 2, NotImplementedError),      # heads must share their options
                             (("neck", "dim_times_res_block_hidden"), 0, NotImplementedError)):
        bad = copy.deepcopy(cfg)
        bad[path[0]][path[1]] = value
        with pytest.raises(exc):
            M(**bad)
    odd = copy.deepcopy(O.named_configs()["moge-2-vits-normal"])            # norm slabs: power-of-two widths 32 ... 1024 only (level 0 is 384 wide here)
    odd["neck"]["res_block_in_norm"] = "layer_norm"; odd["neck"]["num_res_blocks"] = [1, 2, 2, 2, 0]
    with pytest.raises(NotImplementedError):
        M(**odd)
    wide = copy.deepcopy(O.named_configs()["moge-2-vitl-normal"])           # ... and the hidden map counts: 256 x 8 = 2048 is too wide for a hidden norm
    wide["neck"]["res_block_hidden_norm"] = "group_norm"; wide["neck"]["dim_times_res_block_hidden"] = 8
    with pytest.raises(NotImplementedError):
        M(**wide)


def test_v1_mirror_maps_the_head_options():
    """v1.py:62-75: dim_times_res_block_hidden (configs/train/v1.json:31 trains with 2), res_block_norm and the output-block options
    (last_res_blocks, last_conv_size) reach the C ABI; values outside the kernels' range are refused at construction."""
    from moge_amd import _lib as L
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle_v1 as O1
    M1 = import_model_class_by_version("v1")
    rel = M1(**O1.named_configs()["moge-vitl"])._cfg
    assert (rel.hidden_mult, rel.res_block_norm) == (1, L.RES_NORM["group_norm"])
    tr = M1(**O1.named_configs()["moge-vitl-train-config"])._cfg
    assert (tr.hidden_mult, tr.num_res_blocks, list(tr.dim_upsample)[:3]) == (2, 2, [256, 128, 64])
    x4 = M1(**O1.named_configs()["tiny-v1-vits-x4-layer"])._cfg
    assert (x4.hidden_mult, x4.res_block_norm) == (4, L.RES_NORM["layer_norm"])
    base = O1.named_configs()["tiny-v1-vits"]
    last = M1(**O1.named_configs()["tiny-v1-vits-last"])._cfg
    assert (last.last_res_blocks, last.last_conv_size, last.hidden_mult) == (2, 3, 2) and (rel.last_res_blocks, rel.last_conv_size) == (0, 1)
    for key, value in (("last_res_blocks", 9), ("last_conv_size", 5), ("res_block_norm", "instance_norm"), ("dim_times_res_block_hidden", 3),
                       ("dim_times_res_block_hidden", 0)):
        with pytest.raises(NotImplementedError):
            M1(**{**base, key: value})
    with pytest.raises(NotImplementedError):                                  # 512 x 4 = 2048 > 1024: the hidden norm's slab kernels
        M1(**{**O1.named_configs()["moge-vitl"], "dim_upsample": [512, 128, 64], "dim_times_res_block_hidden": 4})
    cfg = L.MogeV1Config()
    cfg.embed_dim, cfg.num_heads, cfg.depth, cfg.n_taps, cfg.dim_proj, cfg.n_up = 384, 6, 12, 1, 128, 1
    cfg.dim_upsample[0], cfg.last_conv_channels, cfg.num_res_blocks, cfg.hidden_mult = 64, 32, 1, 3
    h = ctypes.c_void_p()
    assert L.lib.moge_create_v1(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1 and b"power of two" in L.lib.moge_last_error()


def test_create_rejects_bad_config_without_gpu():
    from moge_amd import _lib as L
    cfg = L.MogeConfig()
    cfg.embed_dim, cfg.num_heads, cfg.depth, cfg.n_taps = 100, 2, 1, 1
    h = ctypes.c_void_p()
    assert L.lib.moge_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1
    assert b"unsupported ViT width" in L.lib.moge_last_error()


def test_eval_plugin_exposes_the_reference_baseline_interface():
    """baselines/moge_mi355x.py is loaded BY PATH by the reference's eval harness and must offer
    Baseline.load (a click command), infer, infer_for_evaluation (moge/test/baseline.py:7-42)."""
    import importlib.util
    import os
    import click
    import torch
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "baselines", "moge_mi355x.py")
    spec = importlib.util.spec_from_file_location("moge_mi355x_plugin", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    B = mod.Baseline
    assert isinstance(B.load, click.Command)
    names = {p.name for p in B.load.params}
    assert names == {"num_tokens", "resolution_level", "pretrained_model_name_or_path", "use_fp16", "device", "version"}       # baselines/moge.py:29-35
    defaults = {p.name: p.default for p in B.load.params}
    assert defaults["version"] == "v1" and defaults["resolution_level"] == 9 and defaults["pretrained_model_name_or_path"] == "Ruicheng/moge-vitl"
    assert defaults["num_tokens"] is None and defaults["device"] == "cuda:0"
    assert callable(B.infer) and callable(B.infer_for_evaluation)
    K = torch.tensor([[[0.8, 0.0, 0.5], [0.0, 1.1, 0.5], [0.0, 0.0, 1.0]]])
    fov = mod._fov_x_degrees(K)
    assert abs(float(fov) - float(torch.rad2deg(2 * torch.atan(torch.tensor(0.5 / 0.8))))) < 1e-5


def test_master_blob_file_header_roundtrip_and_rejects_foreign_files(tmp_path):
    """file format of the packed master blob (host logic only; the GPU round trip is in test_hip_parity.py)."""
    import json
    import numpy as np
    from moge_amd.model.v2 import MoGeModel
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    header = json.dumps({"model_config": cfg, "nbytes": 8192, "layout": "test"}).encode()
    p = tmp_path / "m.blob"
    with open(p, "wb") as f:
        f.write(MoGeModel.BLOB_MAGIC)
        f.write(len(header).to_bytes(8, "little"))
        f.write(header)
        f.write(b"\0" * ((-f.tell()) % 4096))
        np.arange(2048, dtype=np.float32).tofile(f)
    h, off = MoGeModel.read_blob_header(p)
    assert h["nbytes"] == 8192 and off % 4096 == 0 and h["model_config"]["encoder"]["backbone"] == cfg["encoder"]["backbone"]
    m = MoGeModel.from_blob(p)
    assert m._state is None and m._blob_path == str(p) and m.model_config["remap_output"] == cfg["remap_output"]
    with open(p, "ab") as f:
        f.write(b"x")
    with pytest.raises(ValueError):
        MoGeModel.read_blob_header(p)                      # size mismatch = truncated / foreign
    q = tmp_path / "n.blob"
    q.write_bytes(b"not a blob at all")
    with pytest.raises(ValueError):
        MoGeModel.from_blob(q)


def test_master_blob_records_the_model_version(tmp_path):
    """A MoGe-1 blob (same container, moge_amd/model/v1.py) must not load as MoGe-2 and vice versa: read_blob_header raises ValueError, which
    from_pretrained's sidecar path turns into a fall-back to the checkpoint.  Blobs written before the field existed carry no version: their
    header is accepted by either class (a MoGe-1 sidecar of that age must not be rejected on every load) and a config of the other family
    fails in from_blob with the same ValueError."""
    import json
    import numpy as np
    from moge_amd.model.v1 import MoGeModel as V1
    from moge_amd.model.v2 import MoGeModel as V2
    from oracle import moge_oracle as O

    def write(path, header):
        hb = json.dumps(header).encode()
        with open(path, "wb") as f:
            f.write(V2.BLOB_MAGIC)
            f.write(len(hb).to_bytes(8, "little"))
            f.write(hb)
            f.write(b"\0" * ((-f.tell()) % 4096))
            np.zeros(16, dtype=np.float32).tofile(f)

    cfg = O.named_configs()["tiny-vits-normal"]
    p2, p1, p0 = tmp_path / "v2.blob", tmp_path / "v1.blob", tmp_path / "old.blob"
    write(p2, {"model_version": "v2", "model_config": cfg, "nbytes": 64})
    write(p1, {"model_version": "v1", "model_config": {"encoder": "dinov2_vits14"}, "nbytes": 64})
    write(p0, {"model_config": cfg, "nbytes": 64})
    assert V2.read_blob_header(p2)[0]["model_version"] == "v2" and V1.read_blob_header(p1)[0]["model_version"] == "v1"
    V2.read_blob_header(p0); V1.read_blob_header(p0)          # legacy header: no version recorded, either class may try it
    for cls, path in ((V2, p1), (V1, p2)):
        with pytest.raises(ValueError):
            cls.read_blob_header(path)
    with pytest.raises(ValueError):
        V1.from_blob(p0)                                      # ... and a MoGe-2 config does not build a MoGe-1 model
    assert V2.from_blob(p0)._blob_path == str(p0)


def test_v1_rejects_upsample_widths_the_groupnorm_kernels_cannot_run():
    """dim_upsample entries must be 32 / 64 / 128 / 256 / 512 (gn_partial's slabs): rejected at construction, not at the first forward."""
    from moge_amd.model.v1 import MoGeModel as V1
    with pytest.raises(NotImplementedError):
        V1(encoder="dinov2_vits14", dim_upsample=[256, 96, 128])
