"""ctypes binding of libschro_hip.so -- the declarations of include/schro_hip.h.

There is no CPU fallback: if the HIP library is missing or cannot be loaded,
importing the operators raises.  Nothing here touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCHRO_HIP_LIB: another build of the same library (A/B runs of kernel variants on the GPU box)
LIB_PATH = os.environ.get("SCHRO_HIP_LIB") or os.path.join(_HERE, "libschro_hip.so")

# every extern "C" symbol include/schro_hip.h declares
EXPORTED_SYMBOLS = [
    "schro_hip_context_new", "schro_hip_context_free", "schro_hip_device_count",
    "schro_hip_last_error", "schro_hip_set_abort_on_error", "schro_hip_init", "schro_hip_thread_bind", "schro_hip_thread_bound", "schro_hip_context_set_stage_completion", "schro_hip_dequant_plan_new", "schro_hip_dequant_plan_run", "schro_hip_dequant_plan_free", "schro_hip_queue_mark_synchronize",
    "schro_hip_host_alloc", "schro_hip_host_free", "schro_hip_upload_2d_async", "schro_hip_download_2d_async",
    "schro_hip_queue_synchronize", "schro_hip_queue_set_cu_mask", "schro_memory_domain_new_hip_host", "schro_hip_codeblock_layout",
    "schro_frame_to_hip_async", "schro_hipframe_to_cpu_async", "schro_hip_frame_copy_to",
    "schro_upsampled_hipframe_upsample_inplace",
    "schro_hip_scheduler_new_on", "schro_hip_scheduler_publish_reference", "schro_hip_scheduler_reference_frame",
    "schro_hip_scheduler_moves", "schro_hip_scheduler_skipped", "schro_hip_scheduler_refs_in_flight_max",
    "schro_hip_domain_alloc", "schro_hip_domain_free", "schro_hip_domain_bytes",
    "schro_hip_upload_2d", "schro_hip_download_2d", "schro_hip_memset",
    "schro_hip_synchronize", "schro_hip_stream",
    "schro_memory_domain_new_hip", "schro_memory_domain_free_hip", "schro_hip_domain_context",
    "schro_hip_context_domain",
    "schro_hip_obmc_stamps_dump",
    "schro_hip_scheduler_new", "schro_hip_scheduler_new_virtual", "schro_hip_scheduler_free",
    "schro_hip_scheduler_n_devices", "schro_hip_scheduler_context", "schro_hip_scheduler_submit",
    "schro_hip_scheduler_retire", "schro_hip_scheduler_wait",
    "schro_hip_context_select_queue", "schro_hip_context_queue", "schro_hip_queue_wait",
    "schro_hip_queue_mark", "schro_hip_queue_wait_mark",
    "schro_hip_timer_begin", "schro_hip_timer_end",
    "schro_hip_profile_enable", "schro_hip_profile_reset", "schro_hip_profile_read",
    "schro_hip_iiwt_batch", "schro_hip_convert_u8_batch", "schro_hip_upsample_batch",
    "schro_hip_upsampled_bytes", "schro_hip_upsampled_download", "schro_hip_upsampled_pair_bytes",
    "schro_hip_upsampled_pair_download", "schro_hip_pack_u8_batch",
    "schro_hip_pack_v210_batch", "schro_hip_iiwt_pack_v210_batch", "schro_hip_pack_wide_batch", "schro_hip_shift_right_batch",
    "schro_hipframe_shift_right", "schro_hip_add_batch", "schro_hipframe_add",
    "schro_hip_lowdelay_arith", "schro_hip_lowdelay_batch", "schro_hip_dc_predict_batch",
    "schro_hip_dequant_batch",
    "schro_hip_decode_lowdelay_transform_data", "schro_hipframe_dequantise",
    "schro_hip_obmc_batch", "schro_hip_obmc_prediction_epoch", "schro_hip_obmc_overflowed",
    "schro_hip_frame_new_and_alloc", "schro_hip_frame_ref", "schro_hip_frame_unref",
    "schro_frame_to_hip", "schro_hipframe_to_cpu",
    "schro_frame_inverse_iwt_transform_hip", "schro_frame_inverse_iwt_transform_combine_hip", "schro_frame_inverse_iwt_transform_convert_hip", "schro_upsampled_hipframe_upsample",
    "schro_motion_render_hip", "schro_hipframe_convert",
]


class IwtPlane(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_stride", C.c_int),
                ("dst", C.c_void_p), ("dst_stride", C.c_int),
                ("width", C.c_int), ("height", C.c_int),
                ("pred", C.c_void_p), ("pred_stride", C.c_int),
                ("out_width", C.c_int), ("out_height", C.c_int), ("combine", C.c_int),
                ("ll", C.c_void_p), ("ll_stride", C.c_int), ("reserved", C.c_int)]


class ConvertPlane(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_stride", C.c_int),
                ("dst", C.c_void_p), ("dst_stride", C.c_int),
                ("width", C.c_int), ("height", C.c_int)]


class PackPlane(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("src_stride", C.c_int * 3),
                ("src_width", C.c_int), ("src_height", C.c_int),
                ("src_h_shift", C.c_int), ("src_v_shift", C.c_int),
                ("dst", C.c_void_p), ("dst_stride", C.c_int),
                ("width", C.c_int), ("height", C.c_int), ("format", C.c_int)]


class IwtPackPicture(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("src_stride", C.c_int * 3), ("width", C.c_int), ("height", C.c_int),
                ("h_shift", C.c_int), ("v_shift", C.c_int), ("dst", C.c_void_p), ("dst_stride", C.c_int),
                ("out_width", C.c_int), ("out_height", C.c_int)]


class LowDelayParams(C.Structure):
    _fields_ = [("transform_depth", C.c_int),
                ("iwt_luma_width", C.c_int), ("iwt_luma_height", C.c_int),
                ("iwt_chroma_width", C.c_int), ("iwt_chroma_height", C.c_int),
                ("n_horiz_slices", C.c_int), ("n_vert_slices", C.c_int),
                ("slice_bytes_num", C.c_int), ("slice_bytes_denom", C.c_int),
                ("quant_matrix", C.c_int * 19)]


class LowDelayPicture(C.Structure):
    _fields_ = [("slices", C.c_void_p), ("slices_bytes", C.c_size_t),
                ("comp", C.c_void_p * 3), ("stride", C.c_int * 3)]


class Codeblock(C.Structure):
    _fields_ = [("dst_offset", C.c_int), ("dst_stride", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("src_offset", C.c_int), ("src_bytes", C.c_ubyte), ("quant_index", C.c_ubyte),
                ("pad", C.c_ubyte * 2)]


class DequantPlane(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("values", C.c_void_p), ("codeblocks", C.POINTER(Codeblock)),
                ("ncodeblocks", C.c_int), ("is_intra", C.c_int)]


class QuantisedPicture(C.Structure):
    """SchroHipQuantisedPicture (include/schro_hip.h): a picture's codeblock records and quantised values per component."""
    _fields_ = [("codeblocks", C.POINTER(Codeblock) * 3), ("ncodeblocks", C.c_int * 3), ("values", C.c_void_p * 3),
                ("values_bytes", C.c_size_t * 3), ("values_on_device", C.c_int)]


class DcPlane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class UpsamplePlane(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_stride", C.c_int),
                ("dst", C.c_void_p), ("dst_stride", C.c_int),
                ("width", C.c_int), ("height", C.c_int),
                ("src_v", C.c_void_p), ("src_v_stride", C.c_int)]


class ObmcPlane(C.Structure):
    _fields_ = [("mvs", C.c_void_p),
                ("x_num_blocks", C.c_int), ("y_num_blocks", C.c_int),
                ("xblen_luma", C.c_int), ("yblen_luma", C.c_int),
                ("xbsep_luma", C.c_int), ("ybsep_luma", C.c_int),
                ("mv_precision", C.c_int),
                ("picture_weight_bits", C.c_int), ("picture_weight_1", C.c_int),
                ("picture_weight_2", C.c_int),
                ("chroma_h_shift", C.c_int), ("chroma_v_shift", C.c_int),
                ("component", C.c_int),
                ("ref1", C.c_void_p), ("ref1_stride", C.c_int),
                ("ref2", C.c_void_p), ("ref2_stride", C.c_int),
                ("residual", C.c_void_p), ("residual_stride", C.c_int),
                ("residual_bpp", C.c_int),
                ("out", C.c_void_p), ("out_stride", C.c_int),
                ("width", C.c_int), ("height", C.c_int), ("ref_pair", C.c_int), ("prediction_only", C.c_int)]


class FrameData(C.Structure):
    _fields_ = [("format", C.c_int), ("data", C.c_void_p), ("stride", C.c_int),
                ("width", C.c_int), ("height", C.c_int), ("length", C.c_int),
                ("h_shift", C.c_int), ("v_shift", C.c_int)]


class MemoryDomain(C.Structure):
    """The head of SchroHipMemoryDomain == SchroMemoryDomain (schrodomain.h:13-28): the alloc / free table."""
    _fields_ = [("mutex", C.c_void_p), ("flags", C.c_uint),
                ("alloc", C.CFUNCTYPE(C.c_void_p, C.c_int)),
                ("alloc_2d", C.CFUNCTYPE(C.c_void_p, C.c_int, C.c_int, C.c_int)),
                ("free", C.CFUNCTYPE(None, C.c_void_p, C.c_int))]


class Frame(C.Structure):
    """SchroHipFrame == SchroFrame (schroframe.h:69-94), member for member."""


Frame._fields_ = [("refcount", C.c_int), ("free", C.c_void_p), ("domain", C.c_void_p),
                  ("regions", C.c_void_p * 3), ("priv", C.c_void_p),
                  ("format", C.c_int), ("width", C.c_int), ("height", C.c_int),
                  ("components", FrameData * 3),
                  ("is_virtual", C.c_int), ("cached_lines", (C.c_int * 32) * 3),
                  ("virt_frame1", C.POINTER(Frame)), ("virt_frame2", C.POINTER(Frame)),
                  ("render_line", C.c_void_p), ("virt_priv", C.c_void_p), ("virt_priv2", C.c_void_p),
                  ("extension", C.c_int), ("cache_offset", C.c_int * 3),
                  ("is_upsampled", C.c_int), ("upsample_done", C.c_int)]


class GlobalMotion(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("b0", "b1", "a_exp", "a00", "a01", "a10", "a11", "c_exp", "c0", "c1")]


class Params(C.Structure):
    """SchroHipParams == SchroParams (schroparams.h:31-74)."""
    _fields_ = ([("video_format", C.c_void_p), ("is_noarith", C.c_int), ("wavelet_filter_index", C.c_int),
                 ("transform_depth", C.c_int), ("horiz_codeblocks", C.c_int * 7), ("vert_codeblocks", C.c_int * 7)]
                + [(n, C.c_int) for n in ("codeblock_mode_index", "num_refs", "have_global_motion", "xblen_luma",
                                          "yblen_luma", "xbsep_luma", "ybsep_luma", "mv_precision")]
                + [("global_motion", GlobalMotion * 2)]
                + [(n, C.c_int) for n in ("picture_pred_mode", "picture_weight_bits", "picture_weight_1",
                                          "picture_weight_2", "is_lowdelay", "n_horiz_slices", "n_vert_slices",
                                          "slice_bytes_num", "slice_bytes_denom")]
                + [("quant_matrix", C.c_int * 19)]
                + [(n, C.c_int) for n in ("iwt_chroma_width", "iwt_chroma_height", "iwt_luma_width",
                                          "iwt_luma_height", "x_num_blocks", "y_num_blocks", "x_offset", "y_offset")])


class Motion(C.Structure):
    """SchroHipMotion == SchroMotion (schromotion.h:53-86)."""
    _fields_ = ([("src1", C.POINTER(Frame)), ("src2", C.POINTER(Frame)),
                 ("motion_vectors", C.c_void_p), ("params", C.POINTER(Params))]
                + [(n, C.c_int) for n in ("ref_weight_precision", "ref1_weight", "ref2_weight", "mv_precision",
                                          "xoffset", "yoffset", "xbsep", "ybsep", "xblen", "yblen")]
                + [("block", FrameData), ("alloc_block", FrameData), ("obmc_weight", FrameData),
                   ("alloc_block_ref", FrameData * 2), ("block_ref", FrameData * 2),
                   ("weight_x", C.c_int * 64), ("weight_y", C.c_int * 64)]
                + [(n, C.c_int) for n in ("width", "height", "max_fast_x", "max_fast_y", "simple_weight",
                                          "oneref_noscale")])


# int (*SchroHipPictureFunc) (SchroHipContext *ctx, int device_index, void *priv)
PICTURE_FUNC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)

_lib = None


def load():
    """Load libschro_hip.so; raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "schroedinger_amd: %s is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C schroedinger_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    L.schro_hip_context_new.argtypes = [i]
    L.schro_hip_context_new.restype = vp
    L.schro_hip_context_free.argtypes = [vp]
    L.schro_hip_context_free.restype = None
    L.schro_hip_device_count.restype = i
    L.schro_hip_last_error.restype = C.c_char_p
    L.schro_hip_set_abort_on_error.argtypes = [i]
    L.schro_hip_domain_alloc.argtypes = [vp, sz]
    L.schro_hip_domain_alloc.restype = vp
    L.schro_hip_domain_free.argtypes = [vp, vp]
    L.schro_hip_domain_free.restype = i
    L.schro_hip_domain_bytes.argtypes = [vp]
    L.schro_hip_domain_bytes.restype = sz
    L.schro_hip_upload_2d.argtypes = [vp, vp, i, vp, i, i, i]
    L.schro_hip_upload_2d.restype = i
    L.schro_hip_download_2d.argtypes = [vp, vp, i, vp, i, i, i]
    L.schro_hip_download_2d.restype = i
    L.schro_hip_memset.argtypes = [vp, vp, i, sz]
    L.schro_hip_memset.restype = i
    L.schro_hip_synchronize.argtypes = [vp]
    L.schro_hip_synchronize.restype = i
    L.schro_hip_stream.argtypes = [vp]
    L.schro_hip_stream.restype = vp
    L.schro_memory_domain_new_hip.argtypes = [i]
    L.schro_memory_domain_new_hip.restype = vp
    L.schro_memory_domain_free_hip.argtypes = [vp]
    L.schro_memory_domain_free_hip.restype = None
    L.schro_hip_domain_context.argtypes = [vp]
    L.schro_hip_domain_context.restype = vp
    L.schro_hip_context_domain.argtypes = [vp]
    L.schro_hip_context_domain.restype = vp
    L.schro_hip_init.argtypes = []
    L.schro_hip_init.restype = None
    L.schro_hip_thread_bind.argtypes = [vp]
    L.schro_hip_thread_bind.restype = None
    L.schro_hip_thread_bound.argtypes = []
    L.schro_hip_thread_bound.restype = vp
    L.schro_hip_dequant_plan_new.argtypes = [vp, vp, i, i, i]
    L.schro_hip_dequant_plan_new.restype = vp
    L.schro_hip_dequant_plan_run.argtypes = [vp, vp, i]
    L.schro_hip_dequant_plan_run.restype = i
    L.schro_hip_dequant_plan_free.argtypes = [vp]
    L.schro_hip_dequant_plan_free.restype = None
    L.schro_hip_queue_mark_synchronize.argtypes = [vp, i]
    L.schro_hip_queue_mark_synchronize.restype = i
    L.schro_hip_context_set_stage_completion.argtypes = [vp, i]
    L.schro_hip_context_set_stage_completion.restype = i
    L.schro_hip_host_alloc.argtypes = [C.c_size_t]
    L.schro_hip_host_alloc.restype = vp
    L.schro_hip_host_free.argtypes = [vp]
    L.schro_hip_host_free.restype = None
    L.schro_hip_upload_2d_async.argtypes = [vp, vp, i, vp, i, i, i]
    L.schro_hip_upload_2d_async.restype = i
    L.schro_hip_download_2d_async.argtypes = [vp, vp, i, vp, i, i, i]
    L.schro_hip_download_2d_async.restype = i
    L.schro_hip_queue_synchronize.argtypes = [vp, i]
    L.schro_hip_queue_synchronize.restype = i
    L.schro_hip_queue_set_cu_mask.argtypes = [vp, i, C.POINTER(C.c_uint32), i]
    L.schro_hip_queue_set_cu_mask.restype = i
    L.schro_memory_domain_new_hip_host.argtypes = []
    L.schro_memory_domain_new_hip_host.restype = vp
    L.schro_hip_codeblock_layout.argtypes = [i, i, i, C.POINTER(C.c_int), C.POINTER(C.c_int), i, i, C.POINTER(Codeblock), i]
    L.schro_hip_codeblock_layout.restype = i
    L.schro_frame_to_hip_async.argtypes = [vp, vp]
    L.schro_frame_to_hip_async.restype = i
    L.schro_hipframe_to_cpu_async.argtypes = [vp, vp]
    L.schro_hipframe_to_cpu_async.restype = i
    L.schro_hip_frame_copy_to.argtypes = [vp, vp]
    L.schro_hip_frame_copy_to.restype = vp
    L.schro_upsampled_hipframe_upsample_inplace.argtypes = [vp]
    L.schro_upsampled_hipframe_upsample_inplace.restype = i
    L.schro_hip_scheduler_new_on.argtypes = [C.POINTER(C.c_int), i]
    L.schro_hip_scheduler_new_on.restype = vp
    L.schro_hip_scheduler_publish_reference.argtypes = [vp, i, vp]
    L.schro_hip_scheduler_publish_reference.restype = i
    L.schro_hip_scheduler_reference_frame.argtypes = [vp, i, i]
    L.schro_hip_scheduler_reference_frame.restype = vp
    L.schro_hip_scheduler_moves.argtypes = [vp]
    L.schro_hip_scheduler_moves.restype = C.c_long
    L.schro_hip_scheduler_skipped.argtypes = [vp]
    L.schro_hip_scheduler_skipped.restype = C.c_long
    L.schro_hip_scheduler_refs_in_flight_max.argtypes = [vp]
    L.schro_hip_scheduler_refs_in_flight_max.restype = i
    L.schro_hip_scheduler_new.argtypes = [i]
    L.schro_hip_scheduler_new.restype = vp
    L.schro_hip_scheduler_new_virtual.argtypes = [i]
    L.schro_hip_scheduler_new_virtual.restype = vp
    L.schro_hip_scheduler_free.argtypes = [vp]
    L.schro_hip_scheduler_free.restype = None
    L.schro_hip_scheduler_n_devices.argtypes = [vp]
    L.schro_hip_scheduler_n_devices.restype = i
    L.schro_hip_scheduler_context.argtypes = [vp, i]
    L.schro_hip_scheduler_context.restype = vp
    L.schro_hip_scheduler_submit.argtypes = [vp, i, C.POINTER(C.c_int), i, i, PICTURE_FUNC, vp, C.POINTER(C.c_int)]
    L.schro_hip_scheduler_submit.restype = i
    L.schro_hip_scheduler_retire.argtypes = [vp, i]
    L.schro_hip_scheduler_retire.restype = i
    L.schro_hip_scheduler_wait.argtypes = [vp]
    L.schro_hip_scheduler_wait.restype = i
    L.schro_hip_context_select_queue.argtypes = [vp, i]
    L.schro_hip_context_select_queue.restype = i
    L.schro_hip_context_queue.argtypes = [vp]
    L.schro_hip_context_queue.restype = i
    L.schro_hip_queue_wait.argtypes = [vp, i, i]
    L.schro_hip_queue_wait.restype = i
    L.schro_hip_queue_mark.argtypes = [vp, i]
    L.schro_hip_queue_mark.restype = i
    L.schro_hip_queue_wait_mark.argtypes = [vp, i]
    L.schro_hip_queue_wait_mark.restype = i
    L.schro_hip_timer_begin.argtypes = [vp]
    L.schro_hip_timer_begin.restype = i
    L.schro_hip_timer_end.argtypes = [vp]
    L.schro_hip_timer_end.restype = C.c_float
    L.schro_hip_profile_enable.argtypes = [vp, i]
    L.schro_hip_profile_enable.restype = i
    L.schro_hip_profile_reset.argtypes = [vp]
    L.schro_hip_profile_reset.restype = i
    L.schro_hip_profile_read.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.schro_hip_profile_read.restype = i
    L.schro_hip_iiwt_batch.argtypes = [vp, C.POINTER(IwtPlane), i, i, i, i]
    L.schro_hip_iiwt_batch.restype = i
    L.schro_hip_convert_u8_batch.argtypes = [vp, C.POINTER(ConvertPlane), i, i]
    L.schro_hip_convert_u8_batch.restype = i
    L.schro_hip_upsample_batch.argtypes = [vp, C.POINTER(UpsamplePlane), i]
    L.schro_hip_upsample_batch.restype = i
    L.schro_hip_pack_u8_batch.argtypes = [vp, C.POINTER(PackPlane), i]
    L.schro_hip_pack_u8_batch.restype = i
    L.schro_hip_pack_v210_batch.argtypes = [vp, C.POINTER(PackPlane), i, i]
    L.schro_hip_pack_v210_batch.restype = i
    L.schro_hip_iiwt_pack_v210_batch.argtypes = [vp, C.POINTER(IwtPackPicture), i, i, i, i]
    L.schro_hip_iiwt_pack_v210_batch.restype = i
    L.schro_hip_lowdelay_arith.argtypes = [C.POINTER(LowDelayParams), i]
    L.schro_hip_lowdelay_arith.restype = i
    L.schro_hip_lowdelay_batch.argtypes = [vp, C.POINTER(LowDelayPicture), i, C.POINTER(LowDelayParams), i]
    L.schro_hip_lowdelay_batch.restype = i
    L.schro_hip_decode_lowdelay_transform_data.argtypes = [C.POINTER(Frame), vp, C.c_size_t, C.POINTER(LowDelayParams)]
    L.schro_hip_decode_lowdelay_transform_data.restype = i
    L.schro_hip_dc_predict_batch.argtypes = [vp, C.POINTER(DcPlane), i, i]
    L.schro_hip_dc_predict_batch.restype = i
    L.schro_hip_shift_right_batch.argtypes = [vp, C.POINTER(DcPlane), i, i, i]
    L.schro_hip_shift_right_batch.restype = i
    L.schro_hip_pack_wide_batch.argtypes = [vp, C.POINTER(PackPlane), i, i]
    L.schro_hip_pack_wide_batch.restype = i
    L.schro_hipframe_shift_right.argtypes = [C.POINTER(Frame), i]
    L.schro_hipframe_shift_right.restype = i
    L.schro_hip_add_batch.argtypes = [vp, C.POINTER(ConvertPlane), i, i]
    L.schro_hip_add_batch.restype = i
    L.schro_hipframe_add.argtypes = [C.POINTER(Frame), C.POINTER(Frame)]
    L.schro_hipframe_add.restype = i
    L.schro_hip_dequant_batch.argtypes = [vp, C.POINTER(DequantPlane), i, i, i]
    L.schro_hip_dequant_batch.restype = i
    L.schro_hip_upsampled_bytes.argtypes = [i, i, C.POINTER(C.c_int)]
    L.schro_hip_upsampled_bytes.restype = C.c_size_t
    L.schro_hip_upsampled_download.argtypes = [vp, vp, i, vp, i, i, i]
    L.schro_hip_upsampled_download.restype = i
    L.schro_hip_upsampled_pair_bytes.argtypes = [i, i, C.POINTER(C.c_int)]
    L.schro_hip_upsampled_pair_bytes.restype = C.c_size_t
    L.schro_hip_upsampled_pair_download.argtypes = [vp, vp, vp, i, vp, i, i, i]
    L.schro_hip_upsampled_pair_download.restype = i
    L.schro_hip_obmc_batch.argtypes = [vp, C.POINTER(ObmcPlane), i]
    L.schro_hip_obmc_batch.restype = i
    L.schro_hip_obmc_prediction_epoch.argtypes = [vp]
    L.schro_hip_obmc_prediction_epoch.restype = C.c_uint
    L.schro_hip_obmc_overflowed.argtypes = [vp, C.POINTER(C.c_uint), i]
    L.schro_hip_obmc_overflowed.restype = i
    L.schro_hip_frame_new_and_alloc.argtypes = [vp, i, i, i, i]
    L.schro_hip_frame_new_and_alloc.restype = C.POINTER(Frame)
    L.schro_hip_frame_ref.argtypes = [C.POINTER(Frame)]
    L.schro_hip_frame_ref.restype = C.POINTER(Frame)
    L.schro_hip_frame_unref.argtypes = [C.POINTER(Frame)]
    L.schro_hip_frame_unref.restype = None
    L.schro_frame_to_hip.argtypes = [C.POINTER(Frame), C.POINTER(Frame)]
    L.schro_frame_to_hip.restype = i
    L.schro_hipframe_to_cpu.argtypes = [C.POINTER(Frame), C.POINTER(Frame)]
    L.schro_hipframe_to_cpu.restype = i
    L.schro_frame_inverse_iwt_transform_hip.argtypes = [C.POINTER(Frame), C.POINTER(Frame),
                                                        C.POINTER(Params)]
    L.schro_frame_inverse_iwt_transform_hip.restype = i
    L.schro_frame_inverse_iwt_transform_combine_hip.argtypes = [C.POINTER(Frame), C.POINTER(Frame),
                                                                 C.POINTER(Params), C.POINTER(Frame)]
    L.schro_frame_inverse_iwt_transform_combine_hip.restype = i
    L.schro_hipframe_dequantise.argtypes = [C.POINTER(Frame), C.POINTER(QuantisedPicture), C.POINTER(Params)]
    L.schro_hipframe_dequantise.restype = i
    L.schro_frame_inverse_iwt_transform_convert_hip.argtypes = [C.POINTER(Frame), C.POINTER(Frame), C.POINTER(Params)]
    L.schro_frame_inverse_iwt_transform_convert_hip.restype = i
    L.schro_upsampled_hipframe_upsample.argtypes = [C.POINTER(Frame), C.POINTER(Frame)]
    L.schro_upsampled_hipframe_upsample.restype = i
    L.schro_motion_render_hip.argtypes = [C.POINTER(Motion), C.POINTER(Frame), C.POINTER(Frame), i,
                                          C.POINTER(Frame)]
    L.schro_motion_render_hip.restype = i
    L.schro_hipframe_convert.argtypes = [C.POINTER(Frame), C.POINTER(Frame)]
    L.schro_hipframe_convert.restype = i
    _lib = L
    return L


class SchroHipError(RuntimeError):
    code = None         # the SCHRO_HIP_E* value where the error came from a call's return value


ENEEDS_RESIDUAL = -6    # SCHRO_HIP_ENEEDS_RESIDUAL: a routing answer of the combine form, not an error of the stream


def check(rc):
    if rc != 0:
        msg = load().schro_hip_last_error()
        e = SchroHipError("schro_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
        e.code = rc
        raise e
