/* bl_oracle.h - C-ABI of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a plain CPU restatement of the reference's hot-path algorithm
 * (src/geodesic_integrator/*, src/radiation_integrator/{simulation_sampling,
 * simulation_coefficients,formula_coefficients,unpolarized,radiation_geometry}.cpp), written
 * ray-at-a-time so it fits in memory at any camera size. It is pinned against the compiled
 * reference itself (oracle/_ref/blacklight, built by oracle/Makefile from /root/reference/src):
 *   - libbl_oracle_libm.so (built with -DBLO_LIBM, glibc math) must equal the stock reference
 *     bit-for-bit (tier A), and
 *   - libbl_oracle.so (blmath) must equal the reference run under
 *     LD_PRELOAD=oracle/_ref/libblmath_preload.so bit-for-bit (tier B),
 * on the committed golden vectors in tests/golden/ (tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef BL_ORACLE_H_
#define BL_ORACLE_H_

#include "blacklight_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct blo_extra {
  int32_t num_threads;        /* OpenMP threads (<=0: all)                                        */
  /* optional dump of one ray's samples in reference (reversed) order, as in the geodesic
     checkpoint: sample_pos/dir [n][4], sample_len [n]; buffers sized ray_max_steps            */
  int64_t dump_ray;           /* ray index to dump, or -1                                          */
  double *dump_pos, *dump_dir, *dump_len;
  int32_t dump_num;           /* out: samples written                                              */
  /* out: statistics over the call */
  int64_t n_samples, n_gathers, n_flagged;
  int32_t max_sample_num;
  double seconds;             /* wall time of the ray loop                                         */
  /* slow light (slow_light_on; simulation_sampling.cpp:296-349, :736-786, :840-912): in: the time slices the
     reader holds, slow_grids[0] the latest (simulation_reader.cpp:211-303), and the camera time of this
     snapshot; g of blo_render gives the coordinates (= slow_grids[0]'s). out: pixels needing extrapolation
     and by how much: [0] forward small, [1] forward large, [2] backward small, [3] backward large          */
  int32_t slow_n;
  const bl_grid_desc *const *slow_grids;
  const double *slow_times;
  double slow_snapshot_time;
  int64_t slow_count[4];
  double slow_val[4];
  /* in: != 0 renders kappa-distribution electrons in an unpolarized run with kappa_aa_high_i = (3 / kappa)^4.75 + 0.6, the value
     the reference gives it in polarized runs (simulation_coefficients.cpp:121) and leaves uninitialised otherwise (read at :652);
     0: such a run is refused. The library's BL_UNDEFINED_KAPPA. No reference image can pin this configuration. */
  int32_t define_kappa_aa_high_i;
} blo_extra;

/* Same contract as bl_render() with host pointers (d->outputs_on_device must be 0). g may be NULL
 * in formula mode. frame (optional) receives the camera frame. */
int blo_render(const bl_params *p, const bl_grid_desc *g, const bl_render_desc *d,
               bl_camera_frame *frame, double *frequencies, blo_extra *extra, char *err,
               size_t err_len);

/* number of image rows for these parameters (radiation_integrator.cpp:436-520) */
int blo_image_num_quantities(const bl_params *p);
int blo_image_num_frequencies(const bl_params *p);

const char *blo_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
