"""Algorithmic work of one UNet forward: 2 * MACs of every conv / linear / attention matmul of the I2VGen-XL UNet
as the REFERENCE evaluates it (norms, activations, softmax excluded; the reference's per-frame context and
cross-attention K/V projections counted per frame, i.e. without this implementation's dedup).  This is the
per-step figure of SURVEY.md section 8(d) / BASELINE.md section 2 (20.96 TFLOP at B=1, 16 frames, 64x64)."""
from .unet_spec import UNetConfig


def _down(x):
    return (x + 2 - 3) // 2 + 1


def unet_flops(cfg: UNetConfig, B, F, h, w, ntext=77):
    boc, hd, ctxd, ic = cfg.block_out_channels, cfg.attention_head_dim, cfg.cross_attention_dim, cfg.in_channels
    temb = boc[0] * 4
    N = B * F
    total = {"conv3x3": 0.0, "ff": 0.0, "temp_conv": 0.0, "spatial_sdpa": 0.0, "temporal": 0.0, "spatial_proj": 0.0, "other": 0.0}
    P = cfg.context_pool
    nlat = _down(_down(P)) ** 2
    L = ntext + nlat + ic

    def lin(rows, cin, cout):
        return 2.0 * rows * cin * cout

    def conv(rows_out, cin, cout, k=9):
        return 2.0 * rows_out * cin * cout * k

    def resnet(cin, cout, hh, ww):
        r = N * hh * ww
        total["conv3x3"] += conv(r, cin, cout) + conv(r, cout, cout)
        total["other"] += lin(N, temb, cout)
        if cin != cout:
            total["other"] += lin(r, cin, cout)

    def tconv(c, hh, ww):
        total["temp_conv"] += 4 * 2.0 * N * hh * ww * c * c * 3

    def spatial(c, hh, ww):
        r, T = N * hh * ww, hh * ww
        total["spatial_proj"] += 2 * lin(r, c, c)                     # proj_in / proj_out
        total["spatial_proj"] += 4 * lin(r, c, c)                     # self q, k, v, out
        total["spatial_sdpa"] += 4.0 * N * T * T * c                  # QK^T + PV over all heads
        total["spatial_proj"] += 2 * lin(r, c, c) + 2 * lin(N * L, ctxd, c)  # cross q, out; k, v per frame
        total["spatial_sdpa"] += 4.0 * N * T * L * c
        total["ff"] += lin(r, c, 8 * c) + lin(r, 4 * c, c)

    def temporal(cin, inner, hh, ww):
        r = N * hh * ww
        total["temporal"] += 2 * lin(r, cin, inner) + 8 * lin(r, inner, inner) + 2 * 4.0 * r * F * inner
        total["ff"] += lin(r, inner, 8 * inner) + lin(r, 4 * inner, inner)

    # stem
    r0 = N * h * w
    total["other"] += conv(r0, 4, 4 * ic) + conv(r0, 4 * ic, 4 * ic) + conv(r0, 4 * ic, ic)
    total["other"] += conv(N * h * w, 4, 8 * ic) + conv(N * _down(P) ** 2, 8 * ic, 16 * ic) + conv(N * nlat, 16 * ic, ctxd)
    total["other"] += lin(N, ctxd, temb) + lin(N, temb, ctxd * ic) + 2 * (lin(B, boc[0], temb) + lin(B, temb, temb))
    total["other"] += conv(r0, 2 * ic, boc[0])
    temporal(boc[0], cfg.transformer_in_heads * hd, h, w)
    # down
    hh, ww, c_prev = h, w, boc[0]
    skip_c = [boc[0]]
    for i, t in enumerate(cfg.down_block_types):
        c = boc[i]
        for j in range(cfg.layers_per_block):
            resnet(c_prev if j == 0 else c, c, hh, ww)
            tconv(c, hh, ww)
            if t == "CrossAttnDownBlock3D":
                spatial(c, hh, ww)
                temporal(c, c, hh, ww)
            skip_c.append(c)
        if i != len(boc) - 1:
            hh, ww = _down(hh), _down(ww)
            total["conv3x3"] += conv(N * hh * ww, c, c)
            skip_c.append(c)
        c_prev = c
    dims = []  # spatial size per level, for the up path
    hh2, ww2 = h, w
    for i in range(len(boc)):
        dims.append((hh2, ww2))
        if i != len(boc) - 1:
            hh2, ww2 = _down(hh2), _down(ww2)
    c = boc[-1]
    resnet(c, c, hh, ww); tconv(c, hh, ww); spatial(c, hh, ww); temporal(c, c, hh, ww); resnet(c, c, hh, ww); tconv(c, hh, ww)
    rev = list(reversed(boc))
    for i, t in enumerate(cfg.up_block_types):
        c = rev[i]
        hh, ww = dims[len(boc) - 1 - i]
        for j in range(cfg.layers_per_block + 1):
            sc = skip_c.pop()
            resnet(c_prev + sc, c, hh, ww)
            tconv(c, hh, ww)
            if t == "CrossAttnUpBlock3D":
                spatial(c, hh, ww)
                temporal(c, c, hh, ww)
            c_prev = c
        if i != len(boc) - 1:
            nh, nw = dims[len(boc) - 2 - i]
            total["conv3x3"] += conv(N * nh * nw, c, c)
    total["other"] += conv(r0, boc[0], cfg.out_channels)
    total["total"] = sum(total.values())
    return total
