"""Drop-in for the reference's ``i2vgen-xl/pnp_utils.py``: same public names, MI355X engine underneath."""
from mvoc_amd.pnp_utils import (modify_diffuser_attention_forward, register_out_conv_injection,  # noqa: F401
                                register_resnet_injection, register_spatial_attention_pnp, register_temp_attention_pnp,
                                register_temp_conv_injection, register_time, register_time_all)
