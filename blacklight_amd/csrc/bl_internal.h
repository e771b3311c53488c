// bl_internal.h - accessors shared by the host-side translation units of libblacklight_amd.so
// (not part of the C-ABI; hidden visibility).
#ifndef BLACKLIGHT_AMD_BL_INTERNAL_H_
#define BLACKLIGHT_AMD_BL_INTERNAL_H_

#include "../../include/blacklight_amd.h"

const bl_params *bl_internal_params(const bl_ctx *ctx);
const bl_camera_frame *bl_internal_frame(const bl_ctx *ctx);
const double *bl_internal_frequencies(const bl_ctx *ctx, int *count);
int bl_internal_fail(bl_ctx *ctx, int code, const char *message);   // sets bl_last_error, returns code

#endif
