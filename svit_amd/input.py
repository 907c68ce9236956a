"""Decoded uint8 frames as the model input.

The reference's loader normalises on the host and ships fp32 `[B,3,T,S,S]` clips
(slowfast/datasets/ssv2.py:297-327, datasets/utils.py:287-303, utils/misc.py:374-387).  `U8Clips`
keeps what the decoder produced -- uint8 `[V,T,Hs,Ws,3]` -- on the device and describes each
clip as (source video, y0, x0) + size; normalisation, the T H W C -> C T H W permute and the crop
(datasets/transform.py:288-348) happen inside the patch-embedding im2col
(`svit_im2col_patch_u8`).  `model([clips], meta)` accepts it wherever it accepts the fp32 tensor;
the numbers are bit-identical to feeding the reference-normalised fp32 crop.
"""
import torch

from . import hip


def normalize_lut(mean, std, device):
    """bf16 [3,256]: tensor_normalize (datasets/utils.py:287-303) of every uint8 value, with the
    reference's fp32 operation order, then the bf16 rounding the patch-embed operand gets."""
    t = torch.arange(256, dtype=torch.float32) / 255.0
    t = t[None, :] - torch.tensor(list(mean), dtype=torch.float32)[:, None]
    t = t / torch.tensor(list(std), dtype=torch.float32)[:, None]
    return t.to(torch.bfloat16).contiguous().to(device)


class U8Clips:
    """B clips cut from V uint8 videos.  Quacks like the fp32 clip tensor where the model and
    GraphedTrainStep look at it (`shape`, `dim()`, `device`, `detach/clone/contiguous/copy_`)."""

    def __init__(self, frames, size, crops=None, mean=(0.45, 0.45, 0.45), std=(0.225, 0.225, 0.225),
                 lut=None):
        if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [V,T,H,W,3], got %s %s" % (frames.dtype, tuple(frames.shape)))
        if not frames.is_cuda:
            raise hip.SvitHipError("U8Clips lives on the GPU (the host ships uint8, a quarter of the bytes)")
        self.frames = frames.contiguous()
        V, T, Hs, Ws, _ = frames.shape
        self.size = int(size)
        if self.size > Hs or self.size > Ws:
            raise ValueError("crop %d larger than the frames %dx%d" % (self.size, Hs, Ws))
        if crops is None:
            crops = torch.tensor([[v, 0, 0] for v in range(V)], dtype=torch.int32)
        crops = torch.as_tensor(crops, dtype=torch.int32)
        if crops.dim() != 2 or crops.shape[1] != 3:
            raise ValueError("crops must be int32 [B,3] = (video, y0, x0)")
        # range check wherever the table lives (one tiny reduction + one sync at construction; the
        # kernel additionally clamps, so a table rewritten later through copy_ cannot read outside)
        c = crops
        ok = ((c[:, 0] >= 0) & (c[:, 0] < V) & (c[:, 1] >= 0) & (c[:, 1] + self.size <= Hs) &
              (c[:, 2] >= 0) & (c[:, 2] + self.size <= Ws))
        if not bool(ok.all()):
            raise ValueError("crop table outside the frames")
        self.crops = crops.to(frames.device).contiguous()
        self.lut = normalize_lut(mean, std, frames.device) if lut is None else lut

    # ---- the parts of the tensor interface the model path touches ---------------------------
    @property
    def shape(self):
        return torch.Size((self.crops.shape[0], 3, self.frames.shape[1], self.size, self.size))

    @property
    def device(self):
        return self.frames.device

    def dim(self):
        return 5

    def data_ptr(self):
        return self.frames.data_ptr()

    def detach(self):
        return self

    def contiguous(self):
        return self

    def clone(self):
        return U8Clips(self.frames.clone(), self.size, self.crops.clone(), lut=self.lut)

    def copy_(self, other, non_blocking=False):
        self.frames.copy_(other.frames, non_blocking=non_blocking)
        self.crops.copy_(other.crops, non_blocking=non_blocking)
        return self


def spatial_crops_u8(frames, size, num_crops=3, **kw):
    """The test-time crops of every video as a crop TABLE over the shared uint8 frames (no copy):
    clip v*num_crops + j = crop j of video v (transform.uniform_crop offsets)."""
    from .evaluate import uniform_crop_offsets
    V, T, Hs, Ws, _ = frames.shape
    idx = [1] if num_crops == 1 else list(range(num_crops))
    table = [[v, *uniform_crop_offsets(Hs, Ws, size, s)] for v in range(V) for s in idx]
    return U8Clips(frames, size, torch.tensor(table, dtype=torch.int32), **kw)
