"""Drop-in for the reference's ``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py``: the three names the drivers import."""
import numpy as np
import torch

from mvoc_amd.pipeline import I2VGenXLPipeline  # noqa: F401


class I2VGenXLUnetExtension:
    """``composite.py:162-163`` rebinds ``pipe.unet.forward = partial(I2VGenXLUnetExtension.forward, pipe.unet)``"""

    @staticmethod
    @torch.no_grad()
    def forward(model, sample, timestep, fps, image_latents_first, image_latents, image_embeddings=None,
                encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None, multi_frame_guidance=False,
                return_dict=True):
        return model.forward_ext(sample, timestep, fps, image_latents_first, image_latents, image_embeddings,
                                 encoder_hidden_states, multi_frame_guidance=multi_frame_guidance, return_dict=return_dict)


def tensor2vid(video, processor=None, output_type="np"):
    """[B,C,F,H,W] in [-1,1] -> per-batch list of frames (``pipeline_i2vgen_xl.py:82-100``)"""
    outs = []
    for b in range(video.shape[0]):
        v = ((video[b].permute(1, 2, 3, 0).float().cpu() / 2 + 0.5).clamp(0, 1)).numpy()
        if output_type == "pil":
            from PIL import Image
            outs.append([Image.fromarray((f * 255).round().astype("uint8")) for f in v])
        else:
            outs.append(v)
    return np.stack(outs) if output_type == "np" else outs
