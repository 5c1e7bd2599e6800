"""ORACLE (test infrastructure, not product): PnP hooks installed on the oracle UNet.

Equivalent of the reference's ``init_pnp`` (``i2vgen-xl/composite.py:38-60``) + ``register_time_all``
(``i2vgen-xl/pnp_utils.py:48-166``), but state lives in one shared ``PnPState`` instead of ~200 setattr
calls.  Injection sites (``pnp_utils.py:706, 889, 1031, 1099, 1157``): attention Q/K at
``up_blocks[1].{attentions,temp_attentions}[1,2]`` and ``up_blocks[2,3]...[0,1,2]``; features at
``up_blocks[3].resnets[0..2]`` (after conv2), ``up_blocks[3].temp_convs[0..2]`` (after the residual add) and
``conv_out``.  Pinned end-to-end by golden G7 (``tests/test_oracle_golden.py::test_g7_unet_ext_pnp``).
"""
import torch
import torch.nn.functional as F

from . import pnp_ref

ATTN_SITES = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}
FEATURE_BLOCK = 3


class PnPState:
    def __init__(self, conv_schedule=None, spatial_schedule=None, temporal_schedule=None, inject_background=False):
        self.conv_schedule = conv_schedule
        self.spatial_schedule = spatial_schedule
        self.temporal_schedule = temporal_schedule
        self.inject_background = inject_background
        self.t = None
        self.masks = None  # list of (float [1,4,F,h,w], bool [1,4,F,h,w])
        self.ndst = 2      # trailing destination chunks: 2 = [uncond, cond] (reference), 1 = CFG off (generalisation)

    @staticmethod
    def _on(schedule, t):
        return schedule is not None and (t in schedule or t == 1000)

    def conv_on(self):
        return self._on(self.conv_schedule, self.t)

    def spatial_on(self):
        return self._on(self.spatial_schedule, self.t)

    def temporal_on(self):
        return self._on(self.temporal_schedule, self.t)


class _PnPProcessor:
    def __init__(self, state, temporal):
        self.state = state
        self.temporal = temporal

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 height=None, width=None, scale=1.0):
        st = self.state
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k, v = attn.to_k(ctx), attn.to_v(ctx)
        if self.temporal and st.temporal_on():
            q, k = pnp_ref.inject_qk_temporal(q, k, [m[0] for m in st.masks], height, width, st.inject_background, st.ndst)
        elif (not self.temporal) and st.spatial_on():
            nf = b // (len(st.masks) + 1 + st.ndst)
            q, k = pnp_ref.inject_qk_spatial(q, k, [m[1] for m in st.masks], nf, height, width, st.inject_background, st.ndst)
        hd = k.shape[-1] // attn.heads
        q = q.view(b, -1, attn.heads, hd).transpose(1, 2)
        k = k.view(b, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(b, -1, attn.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(b, -1, attn.heads * hd)
        return attn.to_out[1](attn.to_out[0](o))


def install_pnp(unet, state: PnPState):
    for res, blocks in ATTN_SITES.items():
        for i in blocks:
            unet.up_blocks[res].attentions[i].transformer_blocks[0].attn1.processor = _PnPProcessor(state, False)
            unet.up_blocks[res].temp_attentions[i].transformer_blocks[0].attn1.processor = _PnPProcessor(state, True)

    def resnet_forward(rn):
        def forward(input_tensor, temb, scale=1.0):
            h = rn.conv1(rn.nonlinearity(rn.norm1(input_tensor)))
            h = h + rn.time_emb_proj(rn.nonlinearity(temb))[:, :, None, None]
            h = rn.conv2(rn.nonlinearity(rn.norm2(h)))
            if state.conv_on():
                h = pnp_ref.inject_feature_nchw(h, [m[1] for m in state.masks], state.ndst)
            if rn.conv_shortcut is not None:
                input_tensor = rn.conv_shortcut(input_tensor)
            return (input_tensor + h) / rn.output_scale_factor
        return forward

    def tconv_forward(tc, orig):
        def forward(hidden_states, num_frames=1):
            y = orig(hidden_states, num_frames=num_frames)
            if state.conv_on():
                y = pnp_ref.inject_feature_nchw(y, [m[1] for m in state.masks], state.ndst)
            return y
        return forward

    def convout_forward(co, orig):
        def forward(x):
            y = orig(x)
            if state.conv_on():
                y = pnp_ref.inject_feature_nchw(y, [m[1] for m in state.masks], state.ndst)
            return y
        return forward

    blk = unet.up_blocks[FEATURE_BLOCK]
    for i in range(3):
        blk.resnets[i].forward = resnet_forward(blk.resnets[i])
        blk.temp_convs[i].forward = tconv_forward(blk.temp_convs[i], blk.temp_convs[i].forward)
    unet.conv_out.forward = convout_forward(unet.conv_out, unet.conv_out.forward)
    return unet
