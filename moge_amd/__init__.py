"""moge_amd: MI355X-native (gfx950) drop-in for the `moge.model.v2.MoGeModel.infer()` hot path of microsoft/MoGe.

    from moge_amd.model import import_model_class_by_version
    MoGeModel = import_model_class_by_version("v2")
    model = MoGeModel.from_pretrained("model.pt").to("cuda").eval().half()
    out = model.infer(image)        # points, depth, normal, mask, intrinsics - same contract as the reference

Every operation on the path is a hand-written HIP kernel in libmoge_hip.so (see include/moge_hip.h); PyTorch is used
for device memory, the current stream and torch.distributed only."""
__version__ = "0.1.0"
