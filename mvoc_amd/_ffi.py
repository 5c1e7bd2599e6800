"""ctypes binding of libmvoc_hip.so (declared in include/mvoc_hip.h).

The library is the product: importing this module fails loudly when it is missing -- there is no CPU or
PyTorch fallback behind any op in this package.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so; it must be the HIP runtime of the process.  Importing torch first
# makes our DT_NEEDED libamdhip64.so.7 resolve to the already-loaded copy -- loading /opt/rocm's copy first would
# put two HIP runtimes in one process (our launches then fail with "no ROCm-capable device is detected").
import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVOC_HIP_LIB") or os.path.join(_HERE, "libmvoc_hip.so")  # override: A/B builds

A_PLAIN, A_CONV3X3, A_TEMPORAL3 = 0, 1, 2
ACT_NONE, ACT_GEGLU, ACT_SILU, ACT_GELU = 0, 1, 2, 3
FAMILIES = ("gemm", "flash_attn", "temporal_attn", "groupnorm", "layernorm", "pnp", "misc", "temporal_fused")

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [("a", vp), ("a2", vp), ("w", vp), ("out", vp), ("bias", vp), ("rowadd", vp), ("resid", vp),
                ("m", i64), ("n", i64), ("k", i64), ("n_store", i32), ("ldo", i32), ("ldr", i32), ("ld_rowadd", i32),
                ("rowadd_div", i32), ("a_mode", i32), ("lda", i32), ("lda2", i32), ("c1", i32), ("cin", i32),
                ("nimg", i32), ("hout", i32), ("wout", i32), ("hsrc", i32), ("wsrc", i32), ("stride", i32),
                ("upsample", i32), ("hup", i32), ("wup", i32), ("frames", i32), ("hw", i32), ("act", i32), ("tile", i32),
                ("split_k", i32), ("workspace", vp), ("workspace_bytes", sz), ("ln_rowsum", vp), ("ln_bias", vp),
                ("ln_eps", f32), ("pad_mode", i32), ("ln_stats", vp), ("chan_sums", vp), ("row_moments", vp),
                ("row_moments_ld", i32), ("concurrency", i32), ("k_order", i32)]


class AttnDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp),
                ("q_bs", i64), ("q_ts", i64), ("k_bs", i64), ("k_ts", i64), ("v_bs", i64), ("v_ts", i64),
                ("o_bs", i64), ("o_ts", i64),
                ("nbatch", i32), ("heads", i32), ("tq", i32), ("tk", i32), ("kv_bdiv", i32),
                ("head_dim", i32), ("causal", i32), ("scale", C.c_float), ("v2", vp), ("out2", vp), ("pipelined", i32)]


class TAttnDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp),
                ("q_bs", i64), ("q_ps", i64), ("q_ts", i64), ("k_bs", i64), ("k_ps", i64), ("k_ts", i64),
                ("v_bs", i64), ("v_ps", i64), ("v_ts", i64), ("o_bs", i64), ("o_ps", i64), ("o_ts", i64),
                ("nsample", i32), ("hw", i32), ("heads", i32), ("frames", i32)]


class GnDesc(C.Structure):
    _fields_ = [("x", vp), ("x2", vp), ("gamma", vp), ("beta", vp), ("out", vp), ("workspace", vp),
                ("workspace_bytes", sz), ("nsample", i32), ("rows_per_sample", i32), ("c", i32), ("c1", i32),
                ("groups", i32), ("silu", i32), ("eps", f32), ("chan_sums", vp), ("chan_sums2", vp)]


class TFusedDesc(C.Structure):
    _fields_ = [("x", vp), ("wp", vp), ("ln_rowsum", vp), ("ln_bias", vp), ("out", vp), ("nsample", i32), ("frames", i32),
                ("hw", i32), ("c", i32), ("heads", i32), ("ln_eps", f32)]


class XsDesc(C.Structure):
    _fields_ = [("x", vp), ("wp", vp), ("resid", vp), ("out", vp), ("m", i64),
                ("n", i32), ("k", i32), ("n_store", i32), ("ldo", i32), ("ldr", i32), ("act", i32), ("normalize", i32),
                ("ln_eps", C.c_float), ("wp_set_rows", i64)]


class PnpDesc(C.Structure):
    _fields_ = [("x", vp), ("x2", vp), ("masks", vp), ("chunk_stride", i64), ("f_stride", i64), ("p_stride", i64),
                ("nobj", i32), ("frames", i32), ("height", i32), ("width", i32), ("channels", i32), ("mask_h", i32),
                ("mask_w", i32), ("base_chunk0", i32), ("ndst", i32)]


# every symbol include/mvoc_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mvoc_version": (i32, []),
    "mvoc_last_error": (C.c_char_p, []),
    "mvoc_gemm_f16": (i32, [C.POINTER(GemmDesc), vp]),
    "mvoc_gemm_workspace_bytes": (sz, [i64, i64, i64]),
    "mvoc_gemm_chan_sums_written": (i32, []),
    "mvoc_gemm_row_moments_written": (i32, []),
    "mvoc_row_stats_from_moments_f32": (i32, [vp, i64, i32, i32, i32, f32, vp, vp]),
    "mvoc_flash_attn_f16": (i32, [C.POINTER(AttnDesc), vp]),
    "mvoc_temporal_attn_f16": (i32, [C.POINTER(TAttnDesc), vp]),
    "mvoc_temporal_qkv_attn_f16": (i32, [C.POINTER(TFusedDesc), vp]),
    "mvoc_xs_linear_f16": (i32, [C.POINTER(XsDesc), vp]),
    "mvoc_groupnorm_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "mvoc_groupnorm_f16": (i32, [C.POINTER(GnDesc), vp]),
    "mvoc_groupnorm_fold_xs_f16": (i32, [C.POINTER(GnDesc), vp, vp, i32, i32, vp, vp]),
    "mvoc_groupnorm_moments_f16": (i32, [C.POINTER(GnDesc), vp, vp]),
    "mvoc_groupnorm_apply_moments_f16": (i32, [C.POINTER(GnDesc), vp, i32, vp]),
    "mvoc_layernorm_f16": (i32, [vp, vp, vp, vp, i64, i32, f32, vp]),
    "mvoc_row_stats_f16": (i32, [vp, vp, i64, i32, f32, vp]),
    "mvoc_pnp_blend_scatter_tokens": (i32, [C.POINTER(PnpDesc), vp]),
    "mvoc_pnp_blend_scatter_nchw": (i32, [C.POINTER(PnpDesc), vp]),
    "mvoc_ddim_step_f16": (i32, [vp, vp, vp, vp, vp, i64, vp]),
    "mvoc_latent_fusion_f16": (i32, [vp, vp, vp, vp, vp, i32, i64, f64, i32, vp]),
    "mvoc_timestep_embedding_f16": (i32, [vp, i32, i32, vp, vp]),
    "mvoc_act_f16": (i32, [vp, vp, i64, i32, vp]),
    "mvoc_add_f16": (i32, [vp, vp, vp, i64, vp]),
    "mvoc_conv3x3_small_f16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mvoc_adaptive_avgpool_f16": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mvoc_ncfhw_to_tokens_f16": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mvoc_tokens_to_ncfhw_f16": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mvoc_temporal_encoder4_f16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mvoc_conv1x1_small_f16": (i32, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "mvoc_softmax_rows_f16": (i32, [vp, i64, i32, vp]),
    "mvoc_gaussian_sample_f16": (i32, [vp, vp, vp, vp, i64, vp]),
    "mvoc_scale_f16": (i32, [vp, vp, i64, f64, vp]),
    "mvoc_image_to_tokens_f16": (i32, [vp, vp, i32, i32, i32, vp]),
    "mvoc_tokens_to_image_f16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "mvoc_mask_resize_u8": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp]),
    "mvoc_mask_finish": (i32, [vp, vp, vp, i64, vp]),
    "mvoc_permute_rows_f16": (i32, [vp, vp, C.POINTER(i64), C.POINTER(i64), i32, vp]),
    "mvoc_clip_patches_f16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "mvoc_clip_embed_f16": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "mvoc_comm_unique_id": (i32, [vp]),
    "mvoc_comm_init": (i32, [vp, i32, i32, C.POINTER(vp)]),
    "mvoc_comm_destroy": (i32, [vp]),
    "mvoc_allgather_frames": (i32, [vp, vp, vp, sz, vp]),
    "mvoc_alltoall_frames": (i32, [vp, vp, vp, sz, vp]),
    "mvoc_prof_enable": (i32, [i32]),
    "mvoc_prof_collect": (i32, [C.POINTER(f64), C.POINTER(i64), C.POINTER(f64)]),
    "mvoc_prof_reset": (i32, []),
    "mvoc_delay_us": (i32, [i64, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m mvoc_amd.build` (hipcc --offload-arch=gfx950). "
            "mvoc_amd has no CPU / PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


_DEBUG_SYNC = bool(os.environ.get("MVOC_DEBUG_SYNC"))


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libmvoc_hip {what} failed ({rc}): {lib.mvoc_last_error().decode()}")
    if _DEBUG_SYNC:  # debugging aid: localise an asynchronous fault to the op that caused it
        print(f"[mvoc] {what}", flush=True)
        torch.cuda.synchronize()
