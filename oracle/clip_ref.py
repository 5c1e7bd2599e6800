"""ORACLE (test infrastructure, not product): the CLIP towers of MVOC's conditioning prep.

The reference calls transformers' ``CLIPVisionModelWithProjection`` / ``CLIPTextModel``
(``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:739-769`` ``_encode_image``, ``:552-737`` ``encode_prompt``; the reference pins
transformers 4.39 in its requirements).  transformers IS installed in this image (5.x): the oracle is the library's own module
code on CPU in fp32, not a restatement -- parity of the module tree is pinned to that implementation.  What stays
**[recalled]** is the checkpoint's configuration (OpenCLIP ViT-H/14: vision 1280 wide, 32 layers, 16 heads of 80, patch 14,
projection 1024; text 1024 wide, 24 layers, 16 heads, 77 positions, ``hidden_act="gelu"``): ``image_encoder/config.json`` and
``text_encoder/config.json`` are not on disk here.

``vision(cfg_dict)`` / ``text(cfg_dict)`` build the transformers module; ``load(model, state_dict)`` copies a HIP tower's
synthetic weights in (same key names).
"""
import torch


def vision(**cfg):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    d = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224, patch_size=14,
             projection_dim=1024, hidden_act="gelu")
    d.update(cfg)
    return CLIPVisionModelWithProjection(CLIPVisionConfig(**d)).eval().float()


def text(**cfg):
    from transformers import CLIPTextConfig, CLIPTextModel
    d = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, vocab_size=49408,
             max_position_embeddings=77, hidden_act="gelu", bos_token_id=0, eos_token_id=2, pad_token_id=1)
    d.update(cfg)
    return CLIPTextModel(CLIPTextConfig(**d)).eval().float()


def load(model, sd):
    """copy ``sd`` (keys as transformers writes them, with or without the ``text_model.`` prefix) into ``model``"""
    own = model.state_dict()
    out = {}
    for k in own:
        for cand in (k, k[len("text_model."):] if k.startswith("text_model.") else "text_model." + k):
            if cand in sd:
                out[k] = sd[cand].float().cpu()
                break
        else:
            if "position_ids" in k:
                out[k] = own[k]
            else:
                raise KeyError(k)
    model.load_state_dict(out)
    return model


def pixel_values(images, size=224):
    """the reference's host half of ``_encode_image`` after ``_resize_bilinear``: transformers' CLIPImageProcessor with
    do_resize / do_center_crop / do_rescale off and CLIP-statistics normalisation on a float image in [0,1]"""
    import numpy as np
    import PIL.Image
    from transformers import CLIPImageProcessor
    fe = CLIPImageProcessor()
    arr = [np.asarray(im.convert("RGB").resize((size, size), PIL.Image.BILINEAR)).astype(np.float32) / 255.0 for im in images]
    pt = torch.from_numpy(np.stack(arr)).permute(0, 3, 1, 2)
    return fe(images=pt, do_normalize=True, do_center_crop=False, do_resize=False, do_rescale=False, return_tensors="pt").pixel_values
