// bl_host.cpp - the two host-side steps that sit between and after GPU renders, kept on the host as
// in the reference: the adaptive refinement decision (src/radiation_integrator/
// radiation_adaptive.cpp:19-312 + the block bookkeeping of camera.cpp:445-458) and the output writer
// (src/output_writer/output_writer.cpp:169-316, numpy_format.cpp, zip_format.cpp, raw_format.cpp).
// The .npy / ZIP byte layout is the reference's (NumPy .npy v1.0 with a 128-byte header, ZIP 2.0
// stored entries with CRC-32), so its plotting scripts read the files unchanged.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <unistd.h>

#include <fstream>
#include <thread>
#include <string>
#include <vector>

#include "bl_internal.h"
#include "blmath.h"

namespace {

// ------------------------------------------------------------------ adaptive refinement
// EvaluateBlock (radiation_adaptive.cpp:163-312). q(row, col) = intensity of the block.
bool EvaluateBlock(const bl_params &p, const double *block, int bs) {
  auto q_at = [&](int i, int j) { return block[i * bs + j]; };
  auto exceeds = [](int num_exceeded, int num_examined, double frac_cut) {
    double frac = static_cast<double>(num_exceeded) / static_cast<double>(num_examined);
    return frac > frac_cut;
  };
  if (p.adaptive_val_frac >= 0.0) {
    int num_examined = 0, num_exceeded = 0;
    for (int i = 0; i < bs; i++)
      for (int j = 0; j < bs; j++) {
        double q = std::abs(q_at(i, j));
        if (!std::isfinite(q)) continue;
        num_examined++;
        if (q > p.adaptive_val_cut) num_exceeded++;
      }
    if (exceeds(num_exceeded, num_examined, p.adaptive_val_frac)) return true;
  }
  if (p.adaptive_abs_grad_frac >= 0.0) {
    int num_examined = 0, num_exceeded = 0;
    for (int i = 0; i < bs; i++)
      for (int j = 0; j < bs; j++) {
        double q_x = j == 0 ? q_at(i, j + 1) - q_at(i, j)
            : (j == bs - 1 ? q_at(i, j) - q_at(i, j - 1) : 0.5 * (q_at(i, j + 1) - q_at(i, j - 1)));
        double q_y = i == 0 ? q_at(i + 1, j) - q_at(i, j)
            : (i == bs - 1 ? q_at(i, j) - q_at(i - 1, j) : 0.5 * (q_at(i + 1, j) - q_at(i - 1, j)));
        double q = bl_hypot(q_x, q_y);
        if (!std::isfinite(q)) continue;
        num_examined++;
        if (q > p.adaptive_abs_grad_cut) num_exceeded++;
      }
    if (exceeds(num_exceeded, num_examined, p.adaptive_abs_grad_frac)) return true;
  }
  if (p.adaptive_rel_grad_frac >= 0.0) {
    int num_examined = 0, num_exceeded = 0;
    for (int i = 0; i < bs; i++)
      for (int j = 0; j < bs; j++) {
        double q_x;
        if (j == 0)
          q_x = 2.0 * (q_at(i, j + 1) - q_at(i, j)) / (q_at(i, j) + q_at(i, j + 1));
        else if (j == bs - 1)
          q_x = 2.0 * (q_at(i, j) - q_at(i, j - 1)) / (q_at(i, j - 1) + q_at(i, j));
        else
          q_x = 2.0 * (q_at(i, j + 1) - q_at(i, j - 1)) / (q_at(i, j - 1) + 2.0 * q_at(i, j) + q_at(i, j + 1));
        double q_y;
        if (i == 0)
          q_y = 2.0 * (q_at(i + 1, j) - q_at(i, j)) / (q_at(i, j) + q_at(i + 1, j));
        else if (i == bs - 1)
          q_y = 2.0 * (q_at(i, j) - q_at(i - 1, j)) / (q_at(i - 1, j) + q_at(i, j));
        else
          q_y = 2.0 * (q_at(i + 1, j) - q_at(i - 1, j)) / (q_at(i - 1, j) + 2.0 * q_at(i, j) + q_at(i + 1, j));
        double q = bl_hypot(q_x, q_y);
        if (!std::isfinite(q)) continue;
        num_examined++;
        if (q > p.adaptive_rel_grad_cut) num_exceeded++;
      }
    if (exceeds(num_exceeded, num_examined, p.adaptive_rel_grad_frac)) return true;
  }
  if (p.adaptive_abs_lapl_frac >= 0.0) {
    int num_examined = 0, num_exceeded = 0;
    for (int i = 1; i < bs - 1; i++)
      for (int j = 1; j < bs - 1; j++) {
        double q_x = q_at(i, j - 1) - 2.0 * q_at(i, j) + q_at(i, j + 1);
        double q_y = q_at(i - 1, j) - 2.0 * q_at(i, j) + q_at(i + 1, j);
        double q = std::abs(q_x + q_y);
        if (!std::isfinite(q)) continue;
        num_examined++;
        if (q > p.adaptive_abs_lapl_cut) num_exceeded++;
      }
    if (exceeds(num_exceeded, num_examined, p.adaptive_abs_lapl_frac)) return true;
  }
  if (p.adaptive_rel_lapl_frac >= 0.0) {
    int num_examined = 0, num_exceeded = 0;
    for (int i = 1; i < bs - 1; i++)
      for (int j = 1; j < bs - 1; j++) {
        double q_x = 4.0 * (q_at(i, j - 1) - 2.0 * q_at(i, j) + q_at(i, j + 1))
            / (q_at(i, j - 1) + 2.0 * q_at(i, j) + q_at(i, j + 1));
        double q_y = 4.0 * (q_at(i - 1, j) - 2.0 * q_at(i, j) + q_at(i + 1, j))
            / (q_at(i - 1, j) + 2.0 * q_at(i, j) + q_at(i + 1, j));
        double q = std::abs(q_x + q_y);
        if (!std::isfinite(q)) continue;
        num_examined++;
        if (q > p.adaptive_rel_lapl_cut) num_exceeded++;
      }
    if (exceeds(num_exceeded, num_examined, p.adaptive_rel_lapl_frac)) return true;
  }
  return false;
}

// ------------------------------------------------------------------ .npy / ZIP
using Bytes = std::vector<uint8_t>;

// NumPy .npy v1.0 with the reference's fixed 128-byte header (numpy_format.cpp:604-664); empty if the shape does not fit
Bytes MakeNpyHeader(const char *descr, const std::vector<int> &shape) {
  const size_t header_length = 128;
  Bytes out(header_length);
  std::memcpy(out.data(), "\x93NUMPY\x01\x00", 8);
  uint16_t header_len = static_cast<uint16_t>(header_length - 10);
  std::memcpy(out.data() + 8, &header_len, 2);
  std::string dict = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < shape.size(); i++) {
    dict += std::to_string(shape[i]);
    if (shape.size() == 1) dict += ",";
    else if (i + 1 < shape.size()) dict += ", ";
  }
  dict += ")}";
  if (dict.size() > header_length - 11) return Bytes();
  std::memset(out.data() + 10, ' ', header_length - 11);
  std::memcpy(out.data() + 10, dict.data(), dict.size());
  out[header_length - 1] = '\n';
  return out;
}

// One array on its way into a file: the .npy header and the data where they lie (the caller's image rows) or, where the
// reference reorders them (Stokes components, cell averages) or they are locals, a copy. Nothing is assembled in memory:
// a 4096^2 x 64-frequency I_nu record (8.6 GB) is checksummed and written straight from the image.
struct NpyRecord {
  std::string name;
  Bytes head;
  const uint8_t *data = nullptr;
  size_t data_bytes = 0;
  Bytes owned;
  size_t size() const { return head.size() + data_bytes; }
};

// Standard CRC-32 (what zip_format.cpp:284-360 computes with its bit-reversed tables), eight bytes per step
uint32_t Crc32Update(uint32_t crc, const uint8_t *data, size_t n) {
  static uint32_t table[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int t = 1; t < 8; t++) table[t][i] = table[0][table[t - 1][i] & 0xFFu] ^ (table[t - 1][i] >> 8);
    ready = true;
  }
  crc ^= 0xFFFFFFFFu;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, data + i, 4);
    std::memcpy(&hi, data + i + 4, 4);
    lo ^= crc;
    crc = table[7][lo & 0xFFu] ^ table[6][(lo >> 8) & 0xFFu] ^ table[5][(lo >> 16) & 0xFFu] ^ table[4][lo >> 24]
        ^ table[3][hi & 0xFFu] ^ table[2][(hi >> 8) & 0xFFu] ^ table[1][(hi >> 16) & 0xFFu] ^ table[0][hi >> 24];
  }
  for (; i < n; i++) crc = table[0][(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
  return crc ^ 0xFFFFFFFFu;
}

// CRC-32 of the concatenation A | B from crc(A), crc(B) and the length of B: the CRC register is linear over GF(2), so appending
// len_b zero bytes to A is a 32 x 32 bit matrix applied to crc(A) - the matrix for one zero byte squared up along the bits of len_b -
// and crc(A | B) = that ^ crc(B). (The construction zlib's crc32_combine uses; written out here from the algebra.)
uint32_t Gf2Times(const uint32_t *matrix, uint32_t vec) {
  uint32_t sum = 0;
  for (int bit = 0; vec != 0; vec >>= 1, bit++)
    if (vec & 1u) sum ^= matrix[bit];
  return sum;
}
void Gf2Square(uint32_t *square, const uint32_t *matrix) {
  for (int n = 0; n < 32; n++) square[n] = Gf2Times(matrix, matrix[n]);
}
uint32_t Crc32Combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  if (len_b == 0) return crc_a;
  uint32_t even[32], odd[32];
  odd[0] = 0xEDB88320u;   // the register's shift by one zero BIT: the polynomial, then the identity shifted
  for (int n = 1; n < 32; n++) odd[n] = 1u << (n - 1);
  Gf2Square(even, odd);   // two bits
  Gf2Square(odd, even);   // four bits
  do {                     // eight bits = one byte on the first pass, then squared per bit of len_b
    Gf2Square(even, odd);
    if (len_b & 1u) crc_a = Gf2Times(even, crc_a);
    len_b >>= 1;
    if (len_b == 0) break;
    Gf2Square(odd, even);
    if (len_b & 1u) crc_a = Gf2Times(odd, crc_a);
    len_b >>= 1;
  } while (len_b != 0);
  return crc_a ^ crc_b;
}

// Host threads for the byte work on large records (a 4096^2 x 64 image is 8.6 GB: one thread took 3.9 s for its CRC, round 5)
int ByteWorkers(size_t bytes) {
  if (bytes < (32u << 20)) return 1;
  const unsigned int cores = std::thread::hardware_concurrency();
  return static_cast<int>(std::min<size_t>(std::min<unsigned int>(cores == 0 ? 4 : cores, 16), bytes >> 24));
}

// ... of a large array: slices in parallel, combined in order
uint32_t Crc32Large(uint32_t crc, const uint8_t *data, size_t n) {
  const int workers = ByteWorkers(n);
  if (workers <= 1) return Crc32Update(crc, data, n);
  (void)Crc32Update(0, data, 0);   // (the tables, before the threads race to build them)
  const size_t share = (n + workers - 1) / workers;
  std::vector<uint32_t> part(workers, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < workers; t++)
    pool.emplace_back([&, t]() {
      const size_t first = std::min(n, share * t), last = std::min(n, share * (t + 1));
      part[t] = Crc32Update(0, data + first, last - first);
    });
  for (std::thread &t : pool) t.join();
  for (int t = 0; t < workers; t++) {
    const size_t first = std::min(n, share * t), last = std::min(n, share * (t + 1));
    crc = Crc32Combine(crc, part[t], last - first);
  }
  return crc;
}

uint32_t Crc32(const NpyRecord &r) { return Crc32Large(Crc32Update(0, r.head.data(), r.head.size()), r.data, r.data_bytes); }

template <typename T>
void Put(Bytes *b, T v) {
  const uint8_t *p = reinterpret_cast<const uint8_t *>(&v);
  b->insert(b->end(), p, p + sizeof(T));
}

// zip_format.cpp:26-110
// zip64: the record carries its sizes in a ZIP64 extended-information field (APPNOTE 4.5.3) - beyond the reference, which
// stops at 4 GiB (numpy_format.cpp:44-45); used only where 32 bits do not hold the size (or when forced, for the tests)
bool MakeLocalHeader(const std::string &name, const NpyRecord &record, bool zip64, Bytes *header) {
  if (record.size() > UINT32_MAX && !zip64) return false;
  std::string full = name + ".npy";
  header->clear();
  header->insert(header->end(), {0x50, 0x4b, 0x03, 0x04});
  Put<uint8_t>(header, zip64 ? 45 : 20);   // version needed 2.0 (4.5 with ZIP64 fields)
  Put<uint8_t>(header, 0);
  Put<uint16_t>(header, 0);   // flags
  Put<uint16_t>(header, 0);   // stored
  std::time_t now = std::time(nullptr);
  std::tm *lt = std::localtime(&now);
  uint16_t time = static_cast<uint16_t>(lt->tm_hour << 11);
  time |= static_cast<uint16_t>((lt->tm_min & 0x1F) << 5);
  time |= static_cast<uint16_t>(lt->tm_sec / 2 & 0x1F);
  uint16_t date = static_cast<uint16_t>((lt->tm_year - 80) << 9);
  date |= static_cast<uint16_t>(((lt->tm_mon + 1) & 0xF) << 5);
  date |= static_cast<uint16_t>(lt->tm_mday & 0x1F);
  Put<uint16_t>(header, time);
  Put<uint16_t>(header, date);
  Put<uint32_t>(header, Crc32(record));
  Put<uint32_t>(header, zip64 ? 0xFFFFFFFFu : static_cast<uint32_t>(record.size()));
  Put<uint32_t>(header, zip64 ? 0xFFFFFFFFu : static_cast<uint32_t>(record.size()));
  Put<uint16_t>(header, static_cast<uint16_t>(full.size()));
  Put<uint16_t>(header, zip64 ? 20 : 0);
  header->insert(header->end(), full.begin(), full.end());
  if (zip64) {
    Put<uint16_t>(header, 0x0001);
    Put<uint16_t>(header, 16);
    Put<uint64_t>(header, record.size());   // uncompressed, then compressed
    Put<uint64_t>(header, record.size());
  }
  return true;
}

// zip_format.cpp:122-190
Bytes MakeCentralHeader(const Bytes &local, size_t offset, bool zip64, size_t record_size) {
  Bytes out = {0x50, 0x4b, 0x01, 0x02};
  Put<uint8_t>(&out, zip64 ? 45 : 20);   // version made by 2.0 (4.5 with ZIP64 fields)
  Put<uint8_t>(&out, 3);    // unix
  if (!zip64) {
    out.insert(out.end(), local.begin() + 4, local.begin() + 30);
  } else {
    out.insert(out.end(), local.begin() + 4, local.begin() + 28);   // ... up to the name length
    Put<uint16_t>(&out, 28);                                           // extra: sizes and the offset of the local header
  }
  Put<uint16_t>(&out, 0);   // comment length
  Put<uint16_t>(&out, 0);   // disk number
  Put<uint16_t>(&out, 0);   // internal attributes
  Put<uint32_t>(&out, 0x81800000u);
  Put<uint32_t>(&out, zip64 ? 0xFFFFFFFFu : static_cast<uint32_t>(offset));
  uint16_t name_length;
  std::memcpy(&name_length, local.data() + 26, 2);
  out.insert(out.end(), local.begin() + 30, local.begin() + 30 + name_length);
  if (zip64) {
    Put<uint16_t>(&out, 0x0001);
    Put<uint16_t>(&out, 24);
    Put<uint64_t>(&out, record_size);
    Put<uint64_t>(&out, record_size);
    Put<uint64_t>(&out, offset);
  }
  return out;
}

// output_writer.cpp:283-316
bool FormatFilename(const std::string &pattern, int file_number, std::string *out) {
  std::string::size_type open = pattern.find_first_of('{');
  if (open == std::string::npos) return false;
  std::string::size_type close = pattern.find_first_of('}', open);
  if (close == std::string::npos) return false;
  if (pattern[close - 1] != 'd') return false;
  int field_length = 0;
  if (close - open > 2) field_length = std::atoi(pattern.substr(open + 1, close - open - 2).c_str());
  std::string number = std::to_string(file_number);
  std::string zeros;
  if (static_cast<int>(number.size()) < field_length) zeros.assign(field_length - number.size(), '0');
  *out = pattern.substr(0, open) + zeros + number + pattern.substr(close + 1);
  return true;
}

}  // namespace

extern "C" {

int bl_adaptive_refine(const bl_ctx *ctx_const, int level, int n_blocks, const int32_t *block_locs,
                       const double *image, uint8_t *refine_flags, int32_t *n_refined, int32_t *next_locs) {
  bl_ctx *ctx = const_cast<bl_ctx *>(ctx_const);
  if (ctx == nullptr || image == nullptr || refine_flags == nullptr || n_refined == nullptr) return BL_E_ARG;
  const bl_params &p = *bl_internal_params(ctx);
  *n_refined = 0;
  if (p.adaptive_max_level <= 0 || level >= p.adaptive_max_level) {   // radiation_adaptive.cpp:22-23
    for (int b = 0; b < n_blocks; b++) refine_flags[b] = 0;
    return BL_OK;
  }
  const int bs = p.adaptive_block_size;
  const int linear_root_blocks = p.camera_resolution / bs;
  if (level == 0 && n_blocks != linear_root_blocks * linear_root_blocks)
    return bl_internal_fail(ctx, BL_E_ARG, "Level 0 must be evaluated with all root blocks.");
  if (level > 0 && block_locs == nullptr) return bl_internal_fail(ctx, BL_E_ARG, "Refined levels need block_locs.");
  int linear_num_blocks = linear_root_blocks;
  for (int n = 1; n <= level; n++) linear_num_blocks *= 2;
  const int block_num_pix = bs * bs;
  const long long n_pix = level == 0 ? static_cast<long long>(p.camera_resolution) * p.camera_resolution
                                     : static_cast<long long>(n_blocks) * block_num_pix;
  // row of I_nu at the chosen frequency (radiation_adaptive.cpp:166-167): rows 4 l + (I, Q, U, V) when polarized
  const int freq_index = p.image_num_frequencies > 1 ? p.adaptive_frequency_num - 1 : 0;
  const bool polarized = p.model_type == BL_MODEL_SIMULATION && p.image_light && p.has[BL_P_image_polarization] && p.image_polarization;
  const double *row = image + static_cast<size_t>(freq_index) * (polarized ? 4 : 1) * n_pix;
  std::vector<double> block(block_num_pix);
  int count = 0;
  for (int b = 0; b < n_blocks; b++) {
    const int loc_v = level == 0 ? b / linear_root_blocks : block_locs[2 * b + 0];
    const int loc_u = level == 0 ? b % linear_root_blocks : block_locs[2 * b + 1];
    bool flag = false;
    bool decided = false;
    if (p.adaptive_num_regions > 0) {   // forced regions (:52-69, :96-113)
      double y = ((loc_v + 0.5) / linear_num_blocks - 0.5) * p.camera_width;
      double x = ((loc_u + 0.5) / linear_num_blocks - 0.5) * p.camera_width;
      for (int r = 0; r < p.adaptive_num_regions; r++)
        if (level < p.adaptive_region_level[r] && x > p.adaptive_region_x_min[r] && x < p.adaptive_region_x_max[r]
            && y > p.adaptive_region_y_min[r] && y < p.adaptive_region_y_max[r]) {
          flag = true;
          decided = true;
          break;
        }
    }
    if (!decided) {
      if (level == 0) {
        const int j0 = loc_v * bs, i0 = loc_u * bs;
        for (int j = 0; j < bs; j++)
          for (int i = 0; i < bs; i++) block[j * bs + i] = row[static_cast<size_t>(j0 + j) * p.camera_resolution + i0 + i];
      } else {
        std::memcpy(block.data(), row + static_cast<size_t>(b) * block_num_pix, sizeof(double) * block_num_pix);
      }
      flag = EvaluateBlock(p, block.data(), bs);
    }
    refine_flags[b] = flag ? 1 : 0;
    if (flag) {
      if (next_locs != nullptr) {   // camera.cpp:453-458
        int32_t *dst = next_locs + static_cast<size_t>(count) * 8;
        int k = 0;
        for (int v = 2 * loc_v; v <= 2 * loc_v + 1; v++)
          for (int u = 2 * loc_u; u <= 2 * loc_u + 1; u++) {
            dst[2 * k + 0] = v;
            dst[2 * k + 1] = u;
            k++;
          }
      }
      count++;
    }
  }
  *n_refined = count;
  return BL_OK;
}

int bl_write_output(bl_ctx *ctx, const char *path_override, const bl_output_desc *d) {
  if (ctx == nullptr || d == nullptr) return BL_E_ARG;
  if (d->level[0].image == nullptr && bl_image_num_quantities(ctx) > 0) return bl_internal_fail(ctx, BL_E_ARG, "bl_write_output needs the root image.");
  const bl_params &p = *bl_internal_params(ctx);
  const bl_camera_frame &frame = *bl_internal_frame(ctx);
  int n_nu = 0;
  const double *frequencies = bl_internal_frequencies(ctx, &n_nu);
  const int res = p.camera_resolution;
  const int n_q = bl_image_num_quantities(ctx);
  const size_t n_pix = static_cast<size_t>(res) * res;

  std::string path;
  if (path_override != nullptr && path_override[0] != '\0') {
    path = path_override;
  } else {
    if (!p.has[BL_P_output_file]) return bl_internal_fail(ctx, BL_E_MISSING, "OutputWriter unable to find all needed values in input file.");
    path = p.output_file.s;
    if (p.model_type == BL_MODEL_SIMULATION && p.simulation_multiple) {
      int file_number = d->snapshot + (p.slow_light_on ? p.slow_offset : p.simulation_start);
      if (!FormatFilename(p.output_file.s, file_number, &path))
        return bl_internal_fail(ctx, BL_E_INPUT, "Invalid output_file for multiple runs.");
    }
  }
  std::ofstream stream(path, std::ios_base::out | std::ios_base::binary);
  if (!stream.is_open()) return bl_internal_fail(ctx, BL_E_INPUT, "Could not open output file.");

  if (bl_render_num_images(ctx) > 0 && p.output_format != BL_OUTPUT_NPZ)
    return bl_internal_fail(ctx, BL_E_INPUT, "Only npz outputs support rendering.");
  const double *image0 = d->level[0].image;
  if (p.output_format == BL_OUTPUT_RAW && p.model_type == BL_MODEL_SIMULATION && p.image_light && p.has[BL_P_image_polarization]
      && p.image_polarization)   // output_writer.cpp:64-70
    return bl_internal_fail(ctx, BL_E_INPUT, "Only npz or npy outputs support polarization.");
  if (p.output_format == BL_OUTPUT_RAW) {   // raw_format.cpp: the bytes of image[0]
    stream.write(reinterpret_cast<const char *>(image0), static_cast<std::streamsize>(sizeof(double) * n_q * n_pix));
    return BL_OK;
  }
  if (p.output_format == BL_OUTPUT_NPY) {   // numpy_format.cpp:19-32: image[0] as (n_q, res, res)
    Bytes head = MakeNpyHeader("<f8", {n_q, res, res});
    stream.write(reinterpret_cast<const char *>(head.data()), static_cast<std::streamsize>(head.size()));
    stream.write(reinterpret_cast<const char *>(image0), static_cast<std::streamsize>(sizeof(double) * n_q * n_pix));
    return BL_OK;
  }

  // npz (numpy_format.cpp:46-584), records in the reference's order
  std::vector<NpyRecord> records;
  // copy: the data are a local or a reordered temporary; otherwise they stay where the caller holds them
  auto add = [&](const std::string &name, const char *descr, const std::vector<int> &shape, const void *data, size_t bytes, bool copy) {
    NpyRecord r;
    r.name = name;
    r.head = MakeNpyHeader(descr, shape);
    r.data_bytes = bytes;
    if (copy) r.owned.assign(static_cast<const uint8_t *>(data), static_cast<const uint8_t *>(data) + bytes);
    else r.data = static_cast<const uint8_t *>(data);
    records.push_back(std::move(r));
  };
  double mass_msun = frame.mass_msun;
  double width = p.camera_width;
  int32_t num_levels = d->adaptive_num_levels;
  add("mass_msun", "<f8", {1}, &mass_msun, 8, true);
  add("width", "<f8", {1}, &width, 8, true);
  add("frequency", "<f8", {n_nu}, frequencies, sizeof(double) * n_nu, true);
  add("adaptive_num_levels", "<i4", {1}, &num_levels, 4, true);
  if (p.adaptive_max_level > 0) {
    std::vector<int32_t> counts(num_levels + 1);
    counts[0] = (res / p.adaptive_block_size) * (res / p.adaptive_block_size);
    for (int l = 1; l <= num_levels; l++) counts[l] = d->level[l].n_blocks;
    add("adaptive_num_blocks", "<i4", {num_levels + 1}, counts.data(), 4 * counts.size(), true);
  }
  const char *camera_name = p.camera_type == BL_CAMERA_PLANE ? "positions" : "directions";
  if (p.output_camera) {
    if (d->level[0].camera == nullptr) return bl_internal_fail(ctx, BL_E_ARG, "output_camera needs camera data.");
    add(camera_name, "<f8", {res, res, 4}, d->level[0].camera, sizeof(double) * n_pix * 4, false);
  }
  // Stokes rows (numpy_format.cpp:128-165): row (l * stride + a) of the image -> record a, slice l
  const bool polarized = p.model_type == BL_MODEL_SIMULATION && p.image_light && p.has[BL_P_image_polarization] && p.image_polarization;
  const int stokes_stride = polarized ? 4 : 1;
  static const char *const kStokesNames[4] = {"I_nu", "Q_nu", "U_nu", "V_nu"};
  std::vector<double> stokes;
  auto add_stokes = [&](const double *image, size_t pixels, const std::vector<int> &pixel_shape, const std::string &prefix,
                        const std::string &suffix) {
    std::vector<int> shape = pixel_shape;
    if (n_nu > 1) shape.insert(shape.begin(), n_nu);
    if (stokes_stride == 1) {   // unpolarized: the rows are the record
      add(prefix + kStokesNames[0] + suffix, "<f8", shape, image, sizeof(double) * n_nu * pixels, false);
      return;
    }
    for (int a = 0; a < stokes_stride; a++) {
      stokes.resize(static_cast<size_t>(n_nu) * pixels);
      for (int l = 0; l < n_nu; l++)
        std::memcpy(stokes.data() + static_cast<size_t>(l) * pixels, image + static_cast<size_t>(l * stokes_stride + a) * pixels,
                    sizeof(double) * pixels);
      add(prefix + kStokesNames[a] + suffix, "<f8", shape, stokes.data(), sizeof(double) * n_nu * pixels, true);
    }
  };
  if (p.image_light) add_stokes(image0, n_pix, {res, res}, "", "");
  // alternate images (numpy_format.cpp:167-281) and renderings (:282-289), root level
  const bool simulation = p.model_type == BL_MODEL_SIMULATION;
  const int n_render = bl_render_num_images(ctx);
  struct RowSet { const char *name; bool on; int offset; bool per_frequency; bool per_cell; };
  int row = p.image_light ? n_nu * stokes_stride : 0;
  auto take = [&](bool on, int count) { int at = row; if (on) row += count; return at; };
  const int off_time = take(p.image_time, 1), off_length = take(p.image_length, 1), off_lambda = take(p.image_lambda, n_nu);
  const int off_emission = take(p.image_emission, n_nu), off_tau = take(p.image_tau, n_nu);
  const int off_lambda_ave = take(simulation && p.image_lambda_ave, 7 * n_nu);
  const int off_emission_ave = take(simulation && p.image_emission_ave, 7 * n_nu);
  const int off_tau_int = take(simulation && p.image_tau_int, 7 * n_nu);
  const int off_crossings = take(p.image_crossings, 1);
  static const char *const cell_names[7] = {"rho", "n_e", "p_gas", "Theta_e", "B", "sigma", "beta_inverse"};
  // level_shape: the trailing dimensions of one image of the level ({res, res} or {n_blocks, bs, bs})
  auto add_alternates = [&](const double *image, size_t level_pix, const std::vector<int> &level_shape, const std::string &prefix,
                            const std::string &suffix, const double *render) {
    auto shaped = [&](bool per_frequency) {
      std::vector<int> shape;
      if (per_frequency && n_nu > 1) shape.push_back(n_nu);
      shape.insert(shape.end(), level_shape.begin(), level_shape.end());
      return shape;
    };
    auto rows = [&](const std::string &name, int offset, int count, bool per_frequency) {
      add(prefix + name + suffix, "<f8", shaped(per_frequency), image + static_cast<size_t>(offset) * level_pix, sizeof(double) * count * level_pix, false);
    };
    auto cells = [&](const char *stem, int offset) {
      std::vector<double> copy(static_cast<size_t>(n_nu) * level_pix);
      for (int n = 0; n < 7; n++) {
        for (int l = 0; l < n_nu; l++)
          std::memcpy(&copy[static_cast<size_t>(l) * level_pix], image + static_cast<size_t>(offset + l * 7 + n) * level_pix, sizeof(double) * level_pix);
        add(prefix + stem + cell_names[n] + suffix, "<f8", shaped(true), copy.data(), sizeof(double) * copy.size(), true);
      }
    };
    if (p.image_time) rows("time", off_time, 1, false);
    if (p.image_length) rows("length", off_length, 1, false);
    if (p.image_lambda) rows("lambda", off_lambda, n_nu, true);
    if (p.image_emission) rows("emission", off_emission, n_nu, true);
    if (p.image_tau) rows("tau", off_tau, n_nu, true);
    if (simulation && p.image_lambda_ave) cells("lambda_ave_", off_lambda_ave);
    if (simulation && p.image_emission_ave) cells("emission_ave_", off_emission_ave);
    if (simulation && p.image_tau_int) cells("tau_int_", off_tau_int);
    if (p.image_crossings) rows("crossings", off_crossings, 1, false);
    if (n_render > 0) {
      std::vector<int> shape = {n_render, 3};
      shape.insert(shape.end(), level_shape.begin(), level_shape.end());
      add(prefix + "rendering" + suffix, "<f8", shape, render, sizeof(double) * n_render * 3 * level_pix, false);
    }
  };
  if (n_render > 0 && d->level[0].render == nullptr) return bl_internal_fail(ctx, BL_E_ARG, "render_num_images > 0 needs render data.");
  add_alternates(image0, n_pix, {res, res}, "", "", d->level[0].render);
  const int bs = p.adaptive_max_level > 0 ? p.adaptive_block_size : 1;
  for (int l = 1; l <= num_levels; l++) {
    const bl_output_level &lv = d->level[l];
    if (lv.image == nullptr || lv.block_locs == nullptr) return bl_internal_fail(ctx, BL_E_ARG, "Missing adaptive level data.");
    const std::string suffix = "_" + std::to_string(l);
    const size_t level_pix = static_cast<size_t>(lv.n_blocks) * bs * bs;
    add("adaptive_block_locs" + suffix, "<i4", {lv.n_blocks, 2}, lv.block_locs, 8 * static_cast<size_t>(lv.n_blocks), false);
    if (p.output_camera) {
      if (lv.camera == nullptr) return bl_internal_fail(ctx, BL_E_ARG, "output_camera needs camera data.");
      add(std::string("adaptive_") + camera_name + suffix, "<f8", {lv.n_blocks, bs, bs, 4}, lv.camera, sizeof(double) * level_pix * 4, false);
    }
    if (p.image_light) add_stokes(lv.image, level_pix, {lv.n_blocks, bs, bs}, "adaptive_", suffix);
    if (n_render > 0 && lv.render == nullptr) return bl_internal_fail(ctx, BL_E_ARG, "render_num_images > 0 needs render data.");
    add_alternates(lv.image, level_pix, {lv.n_blocks, bs, bs}, "adaptive_", suffix, lv.render);
  }

  // ZIP64 (beyond the reference, whose writer ends with "too large for ZIP" there - numpy_format.cpp:44-45,
  // zip_format.cpp:83-85): records of 4 GiB and more, and records behind the first 4 GiB of the file, carry 64-bit sizes
  // and offsets, and the file ends with the ZIP64 end-of-central-directory record and locator (APPNOTE 4.3.14-15), which
  // numpy.load reads. Files within the 32-bit limits are byte for byte what the reference writes.
  // BLACKLIGHT_AMD_ZIP64 = never keeps the reference's error, = always writes the 64-bit fields whatever the size.
  const char *zip64_env = std::getenv("BLACKLIGHT_AMD_ZIP64");
  const bool zip64_never = zip64_env != nullptr && std::strcmp(zip64_env, "never") == 0;
  const bool zip64_always = zip64_env != nullptr && std::strcmp(zip64_env, "always") == 0;
  std::vector<Bytes> local_headers(records.size()), central_headers(records.size());
  size_t offset = 0;
  bool any_zip64 = zip64_always;
  for (NpyRecord &r : records)
    if (!r.owned.empty()) r.data = r.owned.data();   // (set here: the vector of records has stopped moving)
  for (size_t n = 0; n < records.size(); n++) {
    const size_t size = records[n].size();
    const bool zip64 = !zip64_never && (zip64_always || size > UINT32_MAX || offset > UINT32_MAX);
    if (records[n].head.empty() || !MakeLocalHeader(records[n].name, records[n], zip64, &local_headers[n]))
      return bl_internal_fail(ctx, BL_E_INPUT, "Array and metadata too large for ZIP record.");
    central_headers[n] = MakeCentralHeader(local_headers[n], offset, zip64, size);
    any_zip64 = any_zip64 || zip64;
    offset += local_headers[n].size() + size;
  }
  size_t central_length = 0;
  for (const Bytes &h : central_headers) central_length += h.size();
  const bool end_zip64 = !zip64_never && (any_zip64 || offset > UINT32_MAX || central_length > UINT32_MAX || records.size() > 0xFFFE);
  if (!end_zip64 && (offset > UINT32_MAX || central_length > UINT32_MAX))
    return bl_internal_fail(ctx, BL_E_INPUT, "File contents too large for ZIP format.");
  Bytes end;
  if (end_zip64) {
    end = {0x50, 0x4b, 0x06, 0x06};
    Put<uint64_t>(&end, 44);             // size of the rest of this record
    Put<uint16_t>(&end, 45 | (3 << 8));  // made by 4.5, unix
    Put<uint16_t>(&end, 45);
    Put<uint32_t>(&end, 0);
    Put<uint32_t>(&end, 0);
    Put<uint64_t>(&end, records.size());
    Put<uint64_t>(&end, records.size());
    Put<uint64_t>(&end, central_length);
    Put<uint64_t>(&end, offset);
    end.insert(end.end(), {0x50, 0x4b, 0x06, 0x07});   // locator
    Put<uint32_t>(&end, 0);
    Put<uint64_t>(&end, offset + central_length);
    Put<uint32_t>(&end, 1);
  }
  end.insert(end.end(), {0x50, 0x4b, 0x05, 0x06});
  Put<uint16_t>(&end, 0);
  Put<uint16_t>(&end, 0);
  Put<uint16_t>(&end, static_cast<uint16_t>(std::min<size_t>(records.size(), 0xFFFF)));
  Put<uint16_t>(&end, static_cast<uint16_t>(std::min<size_t>(records.size(), 0xFFFF)));
  Put<uint32_t>(&end, static_cast<uint32_t>(std::min<size_t>(central_length, UINT32_MAX)));
  Put<uint32_t>(&end, static_cast<uint32_t>(std::min<size_t>(offset, UINT32_MAX)));
  Put<uint16_t>(&end, 0);
  // Small pieces through the stream; the data of a large record straight to its place in the file from several host threads (pwrite:
  // the copy into the page cache is what a write costs, and it parallelises) - the same bytes at the same offsets.
  bool written = true;
  size_t position = 0;
  auto put = [&](const uint8_t *bytes, size_t count) {
    stream.write(reinterpret_cast<const char *>(bytes), static_cast<std::streamsize>(count));
    position += count;
  };
  for (size_t n = 0; n < records.size() && written; n++) {
    put(local_headers[n].data(), local_headers[n].size());
    put(records[n].head.data(), records[n].head.size());
    const size_t bytes = records[n].data_bytes;
    const int workers = ByteWorkers(bytes);
    if (workers <= 1) {
      put(records[n].data, bytes);
      continue;
    }
    stream.flush();
    const int fd = ::open(path.c_str(), O_WRONLY);
    if (fd < 0 || !stream.good()) {
      if (fd >= 0) ::close(fd);
      put(records[n].data, bytes);   // (no second descriptor: the plain way)
      continue;
    }
    const size_t share = ((bytes + workers - 1) / workers + 4095) / 4096 * 4096;
    std::vector<char> ok(workers, 1);
    std::vector<std::thread> pool;
    const uint8_t *data = records[n].data;
    for (int t = 0; t < workers; t++)
      pool.emplace_back([&, t]() {
        size_t first = std::min(bytes, share * t);
        const size_t last = std::min(bytes, share * (t + 1));
        while (first < last) {
          const ssize_t did = ::pwrite(fd, data + first, std::min<size_t>(last - first, 256u << 20), static_cast<off_t>(position + first));
          if (did <= 0) {
            ok[t] = 0;
            return;
          }
          first += static_cast<size_t>(did);
        }
      });
    for (std::thread &t : pool) t.join();
    ::close(fd);
    for (char good : ok) written = written && good != 0;
    position += bytes;
    stream.seekp(static_cast<std::streamoff>(position));
  }
  for (const Bytes &h : central_headers) put(h.data(), h.size());
  put(end.data(), end.size());
  return (written && stream.good()) ? BL_OK : bl_internal_fail(ctx, BL_E_INPUT, "Could not write output file.");
}

}  // extern "C"
