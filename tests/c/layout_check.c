/* Member-by-member comparison of include/schro_hip.h's mirror structs with the reference's own
 * structs.  Compiles only where the reference's headers are present (the build container):
 *   gcc -I/root/reference -Iinclude -c tests/c/layout_check.c
 * It is a compile-time check: if it compiles, every assertion holds. */
#define SCHRO_ENABLE_UNSTABLE_API
#include <schroedinger/schroframe.h>
#include <schroedinger/schroparams.h>
#include <schroedinger/schrodomain.h>
#include <stddef.h>
#include "schro_hip.h"

#define SAME(RT, HT, m) _Static_assert (offsetof (RT, m) == offsetof (HT, m) \
    && sizeof (((RT *) 0)->m) == sizeof (((HT *) 0)->m), #RT "." #m)

_Static_assert (sizeof (SchroFrameData) == sizeof (SchroHipFrameData), "SchroFrameData");
SAME (SchroFrameData, SchroHipFrameData, format);
SAME (SchroFrameData, SchroHipFrameData, data);
SAME (SchroFrameData, SchroHipFrameData, stride);
SAME (SchroFrameData, SchroHipFrameData, width);
SAME (SchroFrameData, SchroHipFrameData, height);
SAME (SchroFrameData, SchroHipFrameData, length);
SAME (SchroFrameData, SchroHipFrameData, h_shift);
SAME (SchroFrameData, SchroHipFrameData, v_shift);

_Static_assert (sizeof (SchroFrame) == sizeof (SchroHipFrame), "SchroFrame");
SAME (SchroFrame, SchroHipFrame, refcount);
SAME (SchroFrame, SchroHipFrame, free);
SAME (SchroFrame, SchroHipFrame, domain);
SAME (SchroFrame, SchroHipFrame, regions);
SAME (SchroFrame, SchroHipFrame, priv);
SAME (SchroFrame, SchroHipFrame, format);
SAME (SchroFrame, SchroHipFrame, width);
SAME (SchroFrame, SchroHipFrame, height);
SAME (SchroFrame, SchroHipFrame, components);
SAME (SchroFrame, SchroHipFrame, is_virtual);
SAME (SchroFrame, SchroHipFrame, cached_lines);
SAME (SchroFrame, SchroHipFrame, virt_frame1);
SAME (SchroFrame, SchroHipFrame, virt_frame2);
SAME (SchroFrame, SchroHipFrame, render_line);
SAME (SchroFrame, SchroHipFrame, virt_priv);
SAME (SchroFrame, SchroHipFrame, virt_priv2);
SAME (SchroFrame, SchroHipFrame, extension);
SAME (SchroFrame, SchroHipFrame, cache_offset);
SAME (SchroFrame, SchroHipFrame, is_upsampled);
SAME (SchroFrame, SchroHipFrame, upsample_done);

_Static_assert (sizeof (SchroParams) == sizeof (SchroHipParams), "SchroParams");
SAME (SchroParams, SchroHipParams, video_format);
SAME (SchroParams, SchroHipParams, is_noarith);
SAME (SchroParams, SchroHipParams, wavelet_filter_index);
SAME (SchroParams, SchroHipParams, transform_depth);
SAME (SchroParams, SchroHipParams, horiz_codeblocks);
SAME (SchroParams, SchroHipParams, vert_codeblocks);
SAME (SchroParams, SchroHipParams, codeblock_mode_index);
SAME (SchroParams, SchroHipParams, num_refs);
SAME (SchroParams, SchroHipParams, have_global_motion);
SAME (SchroParams, SchroHipParams, xblen_luma);
SAME (SchroParams, SchroHipParams, yblen_luma);
SAME (SchroParams, SchroHipParams, xbsep_luma);
SAME (SchroParams, SchroHipParams, ybsep_luma);
SAME (SchroParams, SchroHipParams, mv_precision);
SAME (SchroParams, SchroHipParams, global_motion);
SAME (SchroParams, SchroHipParams, picture_pred_mode);
SAME (SchroParams, SchroHipParams, picture_weight_bits);
SAME (SchroParams, SchroHipParams, picture_weight_1);
SAME (SchroParams, SchroHipParams, picture_weight_2);
SAME (SchroParams, SchroHipParams, is_lowdelay);
SAME (SchroParams, SchroHipParams, n_horiz_slices);
SAME (SchroParams, SchroHipParams, n_vert_slices);
SAME (SchroParams, SchroHipParams, slice_bytes_num);
SAME (SchroParams, SchroHipParams, slice_bytes_denom);
SAME (SchroParams, SchroHipParams, quant_matrix);
SAME (SchroParams, SchroHipParams, iwt_chroma_width);
SAME (SchroParams, SchroHipParams, iwt_chroma_height);
SAME (SchroParams, SchroHipParams, iwt_luma_width);
SAME (SchroParams, SchroHipParams, iwt_luma_height);
SAME (SchroParams, SchroHipParams, x_num_blocks);
SAME (SchroParams, SchroHipParams, y_num_blocks);
SAME (SchroParams, SchroHipParams, x_offset);
SAME (SchroParams, SchroHipParams, y_offset);

/* the reference's struct is a prefix of ours (private members follow `slots`) */
_Static_assert (sizeof (SchroMemoryDomain) == offsetof (SchroHipMemoryDomain, ctx), "SchroMemoryDomain");
SAME (SchroMemoryDomain, SchroHipMemoryDomain, mutex);
SAME (SchroMemoryDomain, SchroHipMemoryDomain, flags);
SAME (SchroMemoryDomain, SchroHipMemoryDomain, alloc);
SAME (SchroMemoryDomain, SchroHipMemoryDomain, alloc_2d);
SAME (SchroMemoryDomain, SchroHipMemoryDomain, free);
SAME (SchroMemoryDomain, SchroHipMemoryDomain, slots);
_Static_assert (sizeof (((SchroMemoryDomain *) 0)->slots[0]) == sizeof (((SchroHipMemoryDomain *) 0)->slots[0]), "slot");

/* the frame formats and domain flags this header restates */
_Static_assert (SCHRO_FRAME_FORMAT_v210 == SCHRO_HIP_FORMAT_v210 && SCHRO_FRAME_FORMAT_AYUV == SCHRO_HIP_FORMAT_AYUV
    && SCHRO_FRAME_FORMAT_YUYV == SCHRO_HIP_FORMAT_YUYV && SCHRO_FRAME_FORMAT_UYVY == SCHRO_HIP_FORMAT_UYVY, "formats");
_Static_assert ((SCHRO_MEMORY_DOMAIN_HIP & (SCHRO_MEMORY_DOMAIN_CPU | SCHRO_MEMORY_DOMAIN_CUDA | SCHRO_MEMORY_DOMAIN_OPENGL)) == 0,
    "domain flag is free");
