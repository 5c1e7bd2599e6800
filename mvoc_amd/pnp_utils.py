"""Host-side mirror of the reference's PnP hook layer (``i2vgen-xl/pnp_utils.py``): same function names,
argument meaning, registration sites and per-step state push, so ``composite.init_pnp`` (reference
``i2vgen-xl/composite.py:38-60``) and ``register_time_all`` (``pipeline_i2vgen_xl.py:1684-1685``) work unchanged
against the MI355X engine (``mvoc_amd.unet.I2VGenXLUNet``).

The reference installs Python closures / processor classes that do the masked blend with eager torch ops; here
the hooks only set state on the engine's nodes -- the blend+scatter itself is the HIP kernel
``mvoc_pnp_blend_scatter_tokens`` invoked from the engine's forward at exactly the reference's sites:

* attention Q/K: ``up_blocks[1].{attentions,temp_attentions}[1,2]``, ``up_blocks[2,3]...[0,1,2]``
  (``pnp_utils.py:706, 889``)
* features: ``up_blocks[3].resnets[0..2]`` after conv2 (``:1031``), ``up_blocks[3].temp_convs[0..2]`` (``:1099``),
  ``conv_out`` (``:1157``)
"""
import logging

logger = logging.getLogger(__name__)

ATTN_SITES = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}
FEATURE_BLOCKS = [3]
FEATURE_LAYERS = [0, 1, 2]


def register_time(model, t):
    """legacy helper kept for import compatibility (``pnp_utils.py:36-45``; unused by the reference's loops)"""
    setattr(model.unet.up_blocks[1].resnets[1], "t", t)
    for res in (1, 2, 3):
        for block in (0, 1, 2):
            model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1.processor.t = t
            model.unet.up_blocks[res].temp_attentions[block].transformer_blocks[0].attn1.processor.t = t


def _attn_processors(unet):
    """every processor ``register_time_all`` touches (``pnp_utils.py:64-156``)"""
    for blk in list(unet.down_blocks) + [unet.mid_block] + list(unet.up_blocks):
        for tr in list(blk.attentions) + list(blk.temp_attentions):
            tb = tr.transformer_blocks[0]
            yield tb.attn1.processor
            yield tb.attn2.processor


def register_time_all(model, t, mask):
    """push the current timestep and the list of (float, bool) mask pairs to every hook site
    (``pnp_utils.py:48-166``)"""
    unet = model.unet
    for blk in unet.up_blocks:
        for m in list(blk.resnets) + list(blk.temp_convs):
            m.t, m.mask = t, mask
    for proc in _attn_processors(unet):
        proc.t, proc.mask = t, mask
    for m in (unet.conv_out, unet.conv_in):
        m.t, m.mask = t, mask


def modify_diffuser_attention_forward(unet):
    """The reference rebinds the forwards of TransformerTemporalModel / BasicTransformerBlock / Attention /
    Transformer2DModel only to thread ``height``/``width`` down to the processors (``pnp_utils.py:169-560``).
    The engine's forwards carry the geometry natively; nothing to patch."""
    return unet


def _register_attn(model, injection_schedule, inject_background, temporal):
    for res, blocks in ATTN_SITES.items():
        for block in blocks:
            blk = model.unet.up_blocks[res]
            tr = (blk.temp_attentions if temporal else blk.attentions)[block]
            proc = tr.transformer_blocks[0].attn1.processor
            proc.injection_schedule = injection_schedule
            proc.inject_background = inject_background


def register_spatial_attention_pnp(model, injection_schedule, inject_background=False):
    _register_attn(model, injection_schedule, inject_background, False)


def register_temp_attention_pnp(model, injection_schedule, inject_background=False):
    _register_attn(model, injection_schedule, inject_background, True)


def register_resnet_injection(model, injection_schedule):
    for b in FEATURE_BLOCKS:
        for i in FEATURE_LAYERS:
            model.unet.up_blocks[b].resnets[i].injection_schedule = injection_schedule


def register_temp_conv_injection(model, injection_schedule):
    for b in FEATURE_BLOCKS:
        for i in FEATURE_LAYERS:
            model.unet.up_blocks[b].temp_convs[i].injection_schedule = injection_schedule


def register_out_conv_injection(model, injection_schedule):
    model.unet.conv_out.injection_schedule = injection_schedule
