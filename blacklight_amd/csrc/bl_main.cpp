// bl_main.cpp - command-line driver with the reference's outer contract (src/blacklight.cpp:31-273):
// exactly one argument, the .input file; errors on stdout as "Error: ...\n", warnings on stderr as
// "Warning: ...\n", exit code 0 / 1, the same timing block at the end. Everything between reading the
// parameters and writing the image goes through the C-ABI (include/blacklight_amd.h) to the GPU.
//
// Snapshot input: Athena++ .athdf files (simulation_format = athena, single file or a numbered series
// with simulation_multiple) through bl_snapshot_open(), as the reference's SimulationReader reads them.
// As an extension, a simulation_file that starts with the magic "BLGRID1" (one block) or "BLGRID2"
// (several equal blocks) is taken as a raw grid written by blacklight_amd.mock.Grid.save_raw (dimensions,
// coordinates, primitives: the arrays SimulationReader would have produced), whatever simulation_format says.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "../../include/blacklight_amd.h"

namespace {

double Now() {
  using clock = std::chrono::steady_clock;
  return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}

struct RawGrid {
  int32_t n_b = 1, n_i = 0, n_j = 0, n_k = 0, n_var = 0;
  std::vector<double> coords[6];   // x1f x2f x3f x1v x2v x3v, each [n_b][...]
  std::vector<float> prim;         // [n_var][n_b][n_k][n_j][n_i]
};

// "BLGRID1": int32 n_i, n_j, n_k, n_var (one block); "BLGRID2": int32 n_b, n_i, n_j, n_k, n_var (n_b equal blocks)
bool ReadRawGrid(const std::string &path, RawGrid *g, std::string *error) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) {
    *error = "Could not open file: " + path;
    return false;
  }
  char magic[8] = {};
  f.read(magic, 8);
  const bool v1 = std::memcmp(magic, "BLGRID1\0", 8) == 0, v2 = std::memcmp(magic, "BLGRID2\0", 8) == 0;
  if (!v1 && !v2) {
    *error = "not a raw grid";
    return false;
  }
  if (v2) f.read(reinterpret_cast<char *>(&g->n_b), sizeof(int32_t));
  int32_t dims[4];
  f.read(reinterpret_cast<char *>(dims), sizeof dims);
  g->n_i = dims[0];
  g->n_j = dims[1];
  g->n_k = dims[2];
  g->n_var = dims[3];
  if (!f || g->n_b < 1 || g->n_i < 1 || g->n_j < 1 || g->n_k < 1 || g->n_var < 1) {
    *error = "Could not read file: " + path;
    return false;
  }
  const int counts[6] = {g->n_i + 1, g->n_j + 1, g->n_k + 1, g->n_i, g->n_j, g->n_k};
  for (int c = 0; c < 6; c++) {
    g->coords[c].resize(static_cast<size_t>(g->n_b) * counts[c]);
    f.read(reinterpret_cast<char *>(g->coords[c].data()), sizeof(double) * g->coords[c].size());
  }
  size_t n = static_cast<size_t>(g->n_var) * g->n_b * g->n_k * g->n_j * g->n_i;
  g->prim.resize(n);
  f.read(reinterpret_cast<char *>(g->prim.data()), sizeof(float) * n);
  if (!f) {
    *error = "Could not read file: " + path;
    return false;
  }
  return true;
}

bool IsRawGrid(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  char magic[8] = {};
  f.read(magic, 8);
  return f && (std::memcmp(magic, "BLGRID1\0", 8) == 0 || std::memcmp(magic, "BLGRID2\0", 8) == 0);
}

// ---- several GPUs from one process (BLACKLIGHT_AMD_DEVICES = N or "all"; one host thread per GPU, grid replicated).
// Rays are independent: level 0 is cut into square tiles dealt centre-first round-robin to the devices (the same
// partition as blacklight_amd/distributed.py tile_pixels: cost per ray varies ~4x across the image), refined levels
// give device d the blocks d, d + N, ... of the level's list; every device renders its share through the same
// bl_render call (pixel_map / block_locs) into its own buffer, and the shares are put back in the level's order.

// Pixels (m = m2 * res + m1) of the tiles device `rank` of `world` owns, tile-major, row-major inside a tile
std::vector<int32_t> TilePixels(int res, int rank, int world, int tile) {
  const int per_side = res / tile;
  std::vector<int> ids(static_cast<size_t>(per_side) * per_side);
  for (size_t t = 0; t < ids.size(); t++) ids[t] = static_cast<int>(t);
  const double centre = 0.5 * (per_side - 1);
  auto dist2 = [&](int t) {
    const double dy = t / per_side - centre, dx = t % per_side - centre;
    return dx * dx + dy * dy;
  };
  std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return dist2(a) < dist2(b); });
  std::vector<int32_t> pixels;
  for (size_t t = rank; t < ids.size(); t += world) {
    const int ty = ids[t] / per_side, tx = ids[t] % per_side;
    for (int y = 0; y < tile; y++)
      for (int x = 0; x < tile; x++) pixels.push_back((ty * tile + y) * res + tx * tile + x);
  }
  return pixels;
}

// Largest of 32, 16, 8, ... that divides the camera and is a multiple of the adaptive block
int DefaultTile(int res, int block) {
  int tile = 32;
  while (tile > 1 && (res % tile != 0 || tile % block != 0)) tile /= 2;
  return (res % tile == 0 && tile % block == 0) ? tile : res;
}

// An array of doubles the GPU downloads into at the link's rate: pinned through the library (bl_host_alloc) where that works,
// plain memory otherwise (a host-only context, an allocation the runtime refuses)
struct HostArray {
  bl_ctx *owner = nullptr;
  double *ptr = nullptr;
  size_t count = 0;
  bool pinned = false;
  HostArray() = default;
  HostArray(bl_ctx *ctx, size_t n, bool pin) { Allocate(ctx, n, pin); }
  HostArray(const HostArray &) = delete;
  HostArray &operator=(const HostArray &) = delete;
  HostArray(HostArray &&other) noexcept : owner(other.owner), ptr(other.ptr), count(other.count), pinned(other.pinned) { other.ptr = nullptr; other.count = 0; }
  HostArray &operator=(HostArray &&other) noexcept {
    if (this != &other) {
      Free();
      owner = other.owner; ptr = other.ptr; count = other.count; pinned = other.pinned;
      other.ptr = nullptr;
      other.count = 0;
    }
    return *this;
  }
  ~HostArray() { Free(); }
  void Allocate(bl_ctx *ctx, size_t n, bool pin) {
    Free();
    owner = ctx;
    count = n;
    if (n == 0) return;
    // (pinning costs the runtime a pass over the pages: worth it for buffers that receive several images - the root level's, over a
    // series - not for one download, which four host threads bring in at about the same rate; small arrays never)
    ptr = (pin && n * sizeof(double) >= (8u << 20)) ? static_cast<double *>(bl_host_alloc(ctx, n * sizeof(double))) : nullptr;
    pinned = ptr != nullptr;
    if (!pinned) ptr = static_cast<double *>(std::malloc(n * sizeof(double)));
    if (ptr == nullptr) {
      std::cout << "Error: Could not allocate " << n * sizeof(double) << " bytes for the image.\n";
      std::exit(1);
    }
  }
  void Free() {
    if (ptr != nullptr) {
      if (pinned) bl_host_free(owner, ptr);
      else std::free(ptr);
    }
    ptr = nullptr;
    count = 0;
  }
  double *data() { return ptr; }
  const double *data() const { return ptr; }
};

struct LevelShare {
  std::vector<int32_t> pixels;       // level 0: this device's pixels
  std::vector<int32_t> block_locs;   // refined levels: this device's blocks (v, u)
  std::vector<int64_t> where;        // position of each of its rays in the level's arrays
  HostArray image, camera, render;
  bl_stats stats{};
  int rc = BL_OK;
};

}  // namespace

int main(int argc, char *argv[]) {
  double time_start = Now();
  double time_geodesic = 0.0, time_read = 0.0, time_sample = 0.0, time_image = 0.0, time_render = 0.0;
  if (argc != 2) {
    std::cout << "Error: Must give a single input file.\n";
    return 1;
  }
  static bl_params params;
  char err[1024] = "";
  int num_runs = 1;
  if (bl_params_read_file(&params, argv[1], &num_runs, err, sizeof err) != BL_OK) {
    std::cout << err;
    return 1;
  }
  if (!params.has[BL_P_num_threads]) {
    std::cout << "Error: num_threads not specified in input file.\n";
    return 1;
  }
  // Options that the reference's command line has no place for come from the environment:
  //   BLACKLIGHT_AMD_DEVICES    = N | all   GPUs to spread every image over (default 1; N may exceed the GPUs present:
  //                                         device d % present, which is how one GPU rehearses several)
  //   BLACKLIGHT_AMD_ARITHMETIC = exact | tolerant   the arithmetic tier (read by bl_init itself: tolerant unless it says exact)
  //   BLACKLIGHT_AMD_UNDEFINED  = edge | kappa | edge,kappa   bl_set_undefined_policy(BL_UNDEFINED_EDGE / BL_UNDEFINED_KAPPA)
  int n_devices = 1;
  if (const char *text = std::getenv("BLACKLIGHT_AMD_DEVICES")) {
    const int present = bl_device_count();
    n_devices = std::string(text) == "all" ? present : std::atoi(text);
    if (n_devices < 1 || present < 1) {
      std::cout << "Error: BLACKLIGHT_AMD_DEVICES must be a positive number or \"all\", with a GPU present.\n";
      return 1;
    }
    if (params.has[BL_P_slow_light_on] && params.slow_light_on) n_devices = 1;   // the window of files lives in one context
  }
  std::vector<bl_ctx *> contexts(n_devices, nullptr);
  for (int dev = 0; dev < n_devices; dev++) {
    const int present = n_devices > 1 ? bl_device_count() : 1;
    if (bl_init(&params, n_devices > 1 ? dev % present : -1, &contexts[dev]) != BL_OK) {
      std::cout << bl_last_global_error();
      return 1;
    }
    if (const char *text = std::getenv("BLACKLIGHT_AMD_UNDEFINED")) {
      const std::string chosen(text);
      const int policy = (chosen.find("edge") != std::string::npos ? BL_UNDEFINED_EDGE : 0) | (chosen.find("kappa") != std::string::npos ? BL_UNDEFINED_KAPPA : 0);
      bl_set_undefined_policy(contexts[dev], policy);
    }
    // Adaptive runs decide where to refine from the intensities (radiation_adaptive.cpp:19-312): they are rendered bit-reproducibly
    // in either tier (one transfer record per sample, 1.5 % of a frame), so that two runs of one input take the same decisions -
    // as the reference's do across thread counts. And so are runs over several devices: a frame and its tiles are then the same bits.
    if (params.adaptive_max_level > 0 || n_devices > 1) bl_set_reproducible(contexts[dev], 1);
  }
  bl_ctx *ctx = contexts[0];
  const bool simulation = params.model_type == BL_MODEL_SIMULATION;
  const int res = params.camera_resolution;
  const int n_q = bl_image_num_quantities(ctx);
  const int bs = params.adaptive_max_level > 0 ? params.adaptive_block_size : 1;
  const bool want_camera = params.has[BL_P_output_camera] && params.output_camera && params.output_format == BL_OUTPUT_NPZ;

  HostArray root_image, root_camera, root_render;
  bool have_root = false;
  for (int run = 0; run < num_runs; run++) {
    if (simulation) {
      double t0 = Now();
      RawGrid raw;
      std::string message;
      bl_snapshot *snap = nullptr;
      bl_grid_desc g = {};
      if (params.slow_light_on) {
        // the window of files around this image's camera time (SimulationReader::Read with slow light)
        if (bl_slow_light_read(ctx, run) != BL_OK) {
          std::cout << bl_last_error(ctx);
          return 1;
        }
        std::cerr << bl_warnings(ctx);
        bl_warnings_clear(ctx);
        time_read += Now() - t0;
      } else {
      if (params.has[BL_P_simulation_file] && !params.simulation_multiple && IsRawGrid(params.simulation_file.s)) {
        if (!ReadRawGrid(params.simulation_file.s, &raw, &message)) {
          std::cout << "Error: " << message << "\n";
          return 1;
        }
        g.n_blocks = raw.n_b;
        g.n_i = raw.n_i; g.n_j = raw.n_j; g.n_k = raw.n_k; g.n_var = raw.n_var;
        g.prim = raw.prim.data();
        g.x1f = raw.coords[0].data(); g.x2f = raw.coords[1].data(); g.x3f = raw.coords[2].data();
        g.x1v = raw.coords[3].data(); g.x2v = raw.coords[4].data(); g.x3v = raw.coords[5].data();
        g.ind_rho = 0; g.ind_pgas = 1; g.ind_uu1 = 2; g.ind_uu2 = 3; g.ind_uu3 = 4;
        g.ind_bb1 = 5; g.ind_bb2 = 6; g.ind_bb3 = 7;
        g.ind_kappa = raw.n_var > 8 ? 8 : 0;   // a ninth variable is the electron entropy (plasma_model = code_kappa)
        g.plasma_gamma = params.has[BL_P_plasma_gamma] ? params.plasma_gamma : 0.0;
        g.plasma_gamma_i = params.has[BL_P_plasma_gamma_i] ? params.plasma_gamma_i : 0.0;
        g.plasma_gamma_e = params.has[BL_P_plasma_gamma_e] ? params.plasma_gamma_e : 0.0;
      } else {
        if (bl_snapshot_open(&params, run, &snap, err, sizeof err) != BL_OK) {
          std::cout << err;
          return 1;
        }
        if (run == 0) {   // the reference's order: reader constructor, the integrators' constructors, then the file
          const std::string warnings = bl_snapshot_warnings(snap);
          const size_t setup = bl_snapshot_setup_warning_bytes(snap);
          std::cerr << warnings.substr(0, setup) << bl_warnings(ctx) << warnings.substr(setup);
          bl_warnings_clear(ctx);
        }
        g = *bl_snapshot_grid(snap);
      }
      int rc = BL_OK;
      {   // grid replicated: every device stages its own copy, concurrently
        std::vector<int> results(n_devices, BL_OK);
        std::vector<std::thread> workers;
        for (int dev = 0; dev < n_devices; dev++)
          workers.emplace_back([&, dev]() { results[dev] = bl_set_grid(contexts[dev], &g); });
        for (std::thread &w : workers) w.join();
        for (int dev = 0; dev < n_devices; dev++)
          if (results[dev] != BL_OK && rc == BL_OK) {
            rc = results[dev];
            ctx = contexts[dev];   // whose error text is reported
          }
      }
      bl_snapshot_close(snap);
      if (rc != BL_OK) {
        std::cout << bl_last_error(ctx);
        return 1;
      }
      time_read += Now() - t0;
      }
    }

    // do { Integrate; if (!done) AddGeodesics } while (!done)   (blacklight.cpp:196-233)
    std::vector<HostArray> images, cameras, renders;   // refined levels; the root level's arrays outlive the run (root_image ...)
    const int n_render = bl_render_num_images(ctx);
    std::vector<std::vector<int32_t>> locs(1);
    std::vector<int32_t> counts = {bs > 0 && params.adaptive_max_level > 0 ? (res / bs) * (res / bs) : 0};
    int level = 0;
    std::string shared_warnings;   // several devices: the levels' warnings with their totals
    while (true) {
      const long long n_rays = level == 0 ? static_cast<long long>(res) * res
                                          : static_cast<long long>(counts[level]) * bs * bs;
      if (level == 0 && have_root) {   // the root level's buffers: allocated once, pinned when a series fills them again and again
        images.push_back(std::move(root_image));
        cameras.push_back(std::move(root_camera));
        renders.push_back(std::move(root_render));
      } else {
        const bool pin = level == 0 && num_runs > 1;
        images.emplace_back(ctx, static_cast<size_t>(n_q) * n_rays, pin);
        cameras.emplace_back(ctx, want_camera ? static_cast<size_t>(n_rays) * 4 : 0, pin);
        renders.emplace_back(ctx, static_cast<size_t>(n_render) * 3 * n_rays, pin);
      }
      bl_render_desc d = {};
      d.level = level;
      d.n_blocks = level == 0 ? 0 : counts[level];
      d.block_locs = level == 0 ? nullptr : locs[level].data();
      d.n_rays = n_rays;
      d.image = images.back().data();
      d.render = n_render > 0 ? renders.back().data() : nullptr;
      if (want_camera) {
        if (params.camera_type == BL_CAMERA_PLANE) d.camera_pos = cameras.back().data();
        else d.camera_dir = cameras.back().data();
      }
      double t0 = Now();
      bl_stats st;
      if (n_devices == 1) {
        if (bl_render(ctx, &d) != BL_OK) {
          std::cout << bl_last_error(ctx);
          return 1;
        }
        bl_get_stats(ctx, &st);
      } else {
        // this level's rays over the devices
        const int tile = DefaultTile(res, bs);
        std::vector<LevelShare> shares(n_devices);
        for (int dev = 0; dev < n_devices; dev++) {
          LevelShare &sh = shares[dev];
          if (level == 0) {
            sh.pixels = TilePixels(res, dev, n_devices, tile);
            sh.where.assign(sh.pixels.begin(), sh.pixels.end());
          } else {
            for (int b = dev; b < counts[level]; b += n_devices) {
              sh.block_locs.push_back(locs[level][2 * b]);
              sh.block_locs.push_back(locs[level][2 * b + 1]);
              for (int q = 0; q < bs * bs; q++) sh.where.push_back(static_cast<int64_t>(b) * bs * bs + q);
            }
          }
          const size_t n_local = sh.where.size();
          sh.image.Allocate(contexts[dev], static_cast<size_t>(n_q) * n_local, false);
          sh.camera.Allocate(contexts[dev], want_camera ? n_local * 4 : 0, false);
          sh.render.Allocate(contexts[dev], static_cast<size_t>(n_render) * 3 * n_local, false);
        }
        std::vector<std::thread> workers;
        for (int dev = 0; dev < n_devices; dev++)
          workers.emplace_back([&, dev]() {
            LevelShare &sh = shares[dev];
            if (sh.where.empty()) return;
            bl_render_desc dd = {};
            dd.level = level;
            dd.n_rays = static_cast<int64_t>(sh.where.size());
            if (level == 0) {
              dd.pixel_map = sh.pixels.data();
            } else {
              dd.n_blocks = static_cast<int32_t>(sh.block_locs.size() / 2);
              dd.block_locs = sh.block_locs.data();
            }
            dd.image = sh.image.data();
            dd.render = n_render > 0 ? sh.render.data() : nullptr;
            if (want_camera) {
              if (params.camera_type == BL_CAMERA_PLANE) dd.camera_pos = sh.camera.data();
              else dd.camera_dir = sh.camera.data();
            }
            sh.rc = bl_render(contexts[dev], &dd);
            bl_get_stats(contexts[dev], &sh.stats);
          });
        for (std::thread &w : workers) w.join();
        st = bl_stats{};
        for (int dev = 0; dev < n_devices; dev++) {
          const LevelShare &sh = shares[dev];
          if (sh.rc != BL_OK) {
            std::cout << bl_last_error(contexts[dev]);
            return 1;
          }
          const size_t n_local = sh.where.size();
          // the share's rays back into the level's order, run by run: a tile's row (or a refined block) is consecutive on both sides
          for (size_t i = 0; i < n_local;) {
            const size_t at = static_cast<size_t>(sh.where[i]);
            size_t run = 1;
            while (i + run < n_local && sh.where[i + run] == sh.where[i] + static_cast<int64_t>(run)) run++;
            for (int q = 0; q < n_q; q++)
              std::memcpy(images.back().data() + static_cast<size_t>(q) * n_rays + at, sh.image.data() + static_cast<size_t>(q) * n_local + i, run * sizeof(double));
            for (int q = 0; q < 3 * n_render; q++)
              std::memcpy(renders.back().data() + static_cast<size_t>(q) * n_rays + at, sh.render.data() + static_cast<size_t>(q) * n_local + i, run * sizeof(double));
            if (want_camera) std::memcpy(cameras.back().data() + 4 * at, sh.camera.data() + 4 * i, run * 4 * sizeof(double));
            i += run;
          }
          if (n_local == 0) continue;
          // kernel times add up over devices that worked side by side: keep the slowest device's, sum the counts
          if (sh.stats.ms_total > st.ms_total) {
            st.ms_total = sh.stats.ms_total;
            st.ms_geodesic = sh.stats.ms_geodesic;
            st.ms_locate = sh.stats.ms_locate;
            st.ms_shade = sh.stats.ms_shade;
            st.ms_transfer = sh.stats.ms_transfer;
          }
          st.n_flagged += sh.stats.n_flagged;
          st.n_undefined += sh.stats.n_undefined;
          st.max_sample_num = std::max(st.max_sample_num, sh.stats.max_sample_num);
        }
        // Warnings of the level: the two that carry counts are rebuilt from the devices' totals - the reference's about
        // geodesics that end unexpectedly (geodesics.cpp:389-394) and the library's about samples where the reference reads
        // past its arrays - every other line is taken from whichever devices raised it, once
        std::string others;
        for (int dev = 0; dev < n_devices; dev++) {
          std::istringstream lines(bl_warnings(contexts[dev]));
          std::string line;
          while (std::getline(lines, line)) {
            if (line.find("geodesics terminate unexpectedly") != std::string::npos) continue;
            if (line.find("samples lie where the reference reads past its arrays") != std::string::npos) continue;
            if (others.find(line + "\n") == std::string::npos) others += line + "\n";
          }
          bl_warnings_clear(contexts[dev]);
        }
        if (st.n_flagged > 0)
          shared_warnings += "Warning: " + std::to_string(st.n_flagged) + " out of " + std::to_string(n_rays) + " geodesics terminate unexpectedly.\n";
        if (st.n_undefined > 0)
          shared_warnings += "Warning: " + std::to_string(st.n_undefined) + " samples lie where the reference reads past its arrays; the edge cell was used for them.\n";
        shared_warnings += others;
      }
      double elapsed = Now() - t0;
      // attribute wall time to the reference's three timers in proportion to kernel time
      double kernel = st.ms_total > 0.0f ? st.ms_total : 1.0;
      time_geodesic += elapsed * st.ms_geodesic / kernel;
      // (simulation mode: the locate kernel and half of the coefficient kernel - the grid read - are the reference's sampling stage)
      (simulation ? time_sample : time_image) += elapsed * (st.ms_shade * (simulation ? 0.5 : 1.0) + (simulation ? st.ms_locate : 0.0)) / kernel;
      time_image += elapsed * (st.ms_transfer + (simulation ? 0.5 * st.ms_shade : 0.0)) / kernel;

      if (params.adaptive_max_level <= 0) break;
      double t1 = Now();
      std::vector<uint8_t> flags(counts[level]);
      std::vector<int32_t> next(static_cast<size_t>(counts[level]) * 8);
      int32_t n_refined = 0;
      if (bl_adaptive_refine(ctx, level, counts[level], level == 0 ? nullptr : locs[level].data(),
                             images.back().data(), flags.data(), &n_refined, next.data()) != BL_OK) {
        std::cout << bl_last_error(ctx);
        return 1;
      }
      time_image += Now() - t1;
      if (n_refined == 0) break;
      next.resize(static_cast<size_t>(n_refined) * 8);
      locs.push_back(next);
      counts.push_back(n_refined * 4);
      level++;
    }
    {
      // the reference integrates the geodesics once, before the first image: its count of badly terminated
      // geodesics is reported once, not per image
      std::istringstream lines(std::string(bl_warnings(contexts[0])) + shared_warnings);
      std::string line;
      while (std::getline(lines, line))
        if (run == 0 || line.find("geodesics terminate unexpectedly") == std::string::npos) std::cerr << line << "\n";
      bl_warnings_clear(contexts[0]);
    }

    bl_output_desc out = {};
    out.adaptive_num_levels = level;
    out.snapshot = run;
    for (int l = 0; l <= level; l++) {
      out.level[l].n_blocks = l == 0 ? 0 : counts[l];
      out.level[l].block_locs = l == 0 ? nullptr : locs[l].data();
      out.level[l].image = images[l].data();
      out.level[l].camera = want_camera ? cameras[l].data() : nullptr;
      out.level[l].render = n_render > 0 ? renders[l].data() : nullptr;
    }
    if (bl_write_output(ctx, nullptr, &out) != BL_OK) {
      std::cout << bl_last_error(ctx);
      return 1;
    }
    have_root = true;
    root_image = std::move(images[0]);   // (kept for the next run)
    root_camera = std::move(cameras[0]);
    root_render = std::move(renders[0]);
  }
  root_image.Free();
  root_camera.Free();
  root_render.Free();
  bl_stats last_stats{};
  bl_get_stats(contexts[0], &last_stats);   // (of the last root-level... of the last render of the first device: the tier is the context's)
  for (bl_ctx *c : contexts) bl_free(c);

  double time_full = Now() - time_start;   // blacklight.cpp:259-269
  std::cout << std::setprecision(7);
  std::cout << "\nCalculation completed.";
  std::cout << "\nElapsed time:            " << time_full << " s";
  std::cout << "\n  Integrating geodesics: " << time_geodesic << " s";
  std::cout << "\n  Reading simulation:    " << time_read << " s";
  std::cout << "\n  Sampling simulation:   " << time_sample << " s";
  std::cout << "\n  Integrating image:     " << time_image << " s";
  std::cout << "\n  Rendering:             " << time_render << " s";
  std::cout << "\n\n";
  // (one line more than the reference prints: which arithmetic produced the file. The exact tier's images are the reference's bits -
  // with the pinned math library -, the tolerant tier's lie within north_star's fp64 tolerance of them.)
  std::cout << "blacklight_amd: " << (last_stats.arithmetic == BL_ARITH_TOLERANT ? "tolerant" : "exact") << " arithmetic tier"
            << (last_stats.arithmetic == BL_ARITH_TOLERANT ? (last_stats.composed_maps ? ", composed transfer maps (equal from run to run to rounding)" : ", bit-reproducible") : "")
            << (last_stats.geodesics_reused ? "; geodesics integrated once for the series" : "") << " (BLACKLIGHT_AMD_ARITHMETIC=exact|tolerant)\n";
  return 0;
}
