// bl_internal.h - accessors shared by the host-side translation units of libblacklight_amd.so
// (not part of the C-ABI; hidden visibility).
#ifndef BLACKLIGHT_AMD_BL_INTERNAL_H_
#define BLACKLIGHT_AMD_BL_INTERNAL_H_

#include "../../include/blacklight_amd.h"

// Reader-side bookkeeping of the slow-light window (simulation_reader.cpp:211-303), kept in the context
struct bl_slow_state {
  int first_time = 1;
  int latest_file_number = -1;
  double latest_time = 0.0;     // time[0] of the window
};

const bl_params *bl_internal_params(const bl_ctx *ctx);
bl_slow_state *bl_internal_slow_state(bl_ctx *ctx);
void bl_internal_warn(bl_ctx *ctx, const char *message);   // appends "Warning: ...\n" to bl_warnings
const bl_camera_frame *bl_internal_frame(const bl_ctx *ctx);
const double *bl_internal_frequencies(const bl_ctx *ctx, int *count);
int bl_internal_fail(bl_ctx *ctx, int code, const char *message);   // sets bl_last_error, returns code

#endif
