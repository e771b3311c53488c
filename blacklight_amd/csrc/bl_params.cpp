// bl_params.cpp - host-side .input grammar of the drop-in boundary.
//
// Restates the behaviour of the reference's InputReader::Read()
// (src/input_reader/input_reader.cpp:72-428, enum_readers.cpp:24-233, adaptive_reader.cpp:24-93):
// every whitespace character is stripped, '#' starts a comment, each remaining line must be
// key=value, unknown keys are fatal, booleans are exactly true/false, angles are given in degrees,
// numbers are parsed with std::stod / std::stoi / std::stof semantics (leading number, trailing
// text ignored). Error texts are the reference's.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "../../include/blacklight_amd.h"
#include "blmath.h"

namespace {

constexpr double kPi = 3.141592653589793;  // reference src/blacklight.hpp:12

struct FieldInfo {
  const char *name;
  char kind;
  size_t offset;
};

#define BL_X_INFO(kind, name) {#name, #kind[0], offsetof(bl_params, name)},
const FieldInfo kFields[] = {BL_PARAM_LIST(BL_X_INFO)};
#undef BL_X_INFO

// Fields that exist in the block but are not keys of the grammar (they are derived or set through
// a compound key): camera_pole <- camera_th; cut_plane_{origin,normal}_{x,y,z} <- triples.
bool IsDerivedField(int index) {
  return index == BL_P_camera_pole || (index >= BL_P_cut_plane_origin_x && index <= BL_P_cut_plane_normal_z);
}

void SetError(char *err, size_t err_len, const std::string &message) {
  if (err != nullptr && err_len > 0) std::snprintf(err, err_len, "Error: %s\n", message.c_str());
}

struct ParseFailure {
  std::string message;
};

// std::stod / std::stoi / std::stof of the reference: a failed conversion throws
// std::invalid_argument, which main() reports as "Could not read input file."
// (src/blacklight.cpp:69-72).
double ToDouble(const std::string &val, size_t *consumed = nullptr) {
  const char *begin = val.c_str();
  char *end = nullptr;
  double x = std::strtod(begin, &end);
  if (end == begin) throw ParseFailure{"Could not read input file."};
  if (consumed != nullptr) *consumed = static_cast<size_t>(end - begin);
  return x;
}
int ToInt(const std::string &val, size_t *consumed = nullptr) {
  const char *begin = val.c_str();
  char *end = nullptr;
  long x = std::strtol(begin, &end, 10);
  if (end == begin) throw ParseFailure{"Could not read input file."};
  if (consumed != nullptr) *consumed = static_cast<size_t>(end - begin);
  return static_cast<int>(x);
}
float ToFloat(const std::string &val) {
  const char *begin = val.c_str();
  char *end = nullptr;
  float x = std::strtof(begin, &end);
  if (end == begin) throw ParseFailure{"Could not read input file."};
  return x;
}
bool ToBool(const std::string &val) {  // input_reader.cpp:451-459
  if (val == "true") return true;
  if (val == "false") return false;
  throw ParseFailure{"Unknown string used for boolean value."};
}

int ToEnum(int index, const std::string &val) {  // enum_readers.cpp
  struct Choice { const char *text; int value; };
  auto pick = [&](std::initializer_list<Choice> choices, const char *type_name) -> int {
    for (const Choice &c : choices)
      if (val == c.text) return c.value;
    throw ParseFailure{std::string("Unknown string used for ") + type_name + " value."};
  };
  switch (index) {
    case BL_P_model_type:
      return pick({{"simulation", BL_MODEL_SIMULATION}, {"formula", BL_MODEL_FORMULA}}, "ModelType");
    case BL_P_output_format:
      return pick({{"npz", BL_OUTPUT_NPZ}, {"npy", BL_OUTPUT_NPY}, {"raw", BL_OUTPUT_RAW}}, "OutputFormat");
    case BL_P_simulation_format:
      return pick({{"athena", BL_SIMFMT_ATHENA}, {"athenak", BL_SIMFMT_ATHENAK},
                   {"iharm3d", BL_SIMFMT_IHARM3D}, {"harm3d", BL_SIMFMT_HARM3D}}, "SimulationFormat");
    case BL_P_simulation_coord:
      return pick({{"cks", BL_COORD_CKS}, {"sks", BL_COORD_SKS}, {"mks", BL_COORD_SKS},
                   {"fmks", BL_COORD_FMKS}}, "Coordinates");
    case BL_P_camera_type:
      return pick({{"plane", BL_CAMERA_PLANE}, {"pinhole", BL_CAMERA_PINHOLE}}, "Camera");
    case BL_P_ray_terminate:
      return pick({{"photon", BL_TERMINATE_PHOTON}, {"multiplicative", BL_TERMINATE_MULTIPLICATIVE},
                   {"additive", BL_TERMINATE_ADDITIVE}}, "RayTerminate");
    case BL_P_ray_integrator:
      return pick({{"dp", BL_INTEGRATOR_DP}, {"rk4", BL_INTEGRATOR_RK4}, {"rk2", BL_INTEGRATOR_RK2}},
                  "RayIntegrator");
    case BL_P_image_frequency_spacing:
      return pick({{"lin_freq", BL_SPACING_LIN_FREQ}, {"lin_wave", BL_SPACING_LIN_WAVE},
                   {"log", BL_SPACING_LOG}}, "FrequencySpacing");
    case BL_P_image_normalization:
      return pick({{"camera", BL_NORM_CAMERA}, {"infinity", BL_NORM_INFINITY}}, "FrequencyNormalization");
    case BL_P_plasma_model:
      return pick({{"ti_te_beta", BL_PLASMA_TI_TE_BETA}, {"code_kappa", BL_PLASMA_CODE_KAPPA}}, "PlasmaModel");
    default:
      throw ParseFailure{"Could not read input file."};
  }
}

template <typename T>
T *FieldPtr(bl_params *p, const FieldInfo &f) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(p) + f.offset);
}
template <typename T>
const T *FieldPtr(const bl_params *p, const FieldInfo &f) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(p) + f.offset);
}

void SetDouble(bl_params *p, int index, double v) {
  *FieldPtr<double>(p, kFields[index]) = v;
  p->has[index] = 1;
}

// "x,y,z" (input_reader.cpp:468-482)
void ReadTriple(const std::string &val, bl_params *p, int first_index) {
  size_t pos_1 = 0, pos_2 = 0;
  double x = ToDouble(val, &pos_1);
  if (pos_1 + 1 > val.size()) throw ParseFailure{"Could not read input file."};
  std::string rest_1 = val.substr(pos_1 + 1);
  double y = ToDouble(rest_1, &pos_2);
  if (pos_1 + pos_2 + 2 > val.size()) throw ParseFailure{"Could not read input file."};
  double z = ToDouble(val.substr(pos_1 + pos_2 + 2));
  if (val[pos_1] != ',' || val[pos_1 + pos_2 + 1] != ',')
    throw ParseFailure{"Invalid triple (" + val + ") in input file."};
  SetDouble(p, first_index, x);
  SetDouble(p, first_index + 1, y);
  SetDouble(p, first_index + 2, z);
}

// adaptive_num_regions / adaptive_region_<n>_<field> (adaptive_reader.cpp:24-93). Regions beyond
// adaptive_num_regions are silently ignored, as there; regions beyond BL_MAX_REGIONS are an error
// of this build.
void ReadAdaptiveRegion(bl_params *p, const std::string &key, const std::string &val) {
  static const char *const suffixes[5] = {"_level", "_x_min", "_x_max", "_y_min", "_y_max"};
  for (int s = 0; s < 5; s++) {
    if (key.size() >= 7 && key.compare(key.size() - 6, std::string::npos, suffixes[s]) == 0) {
      int region = ToInt(key.substr(0, key.size() - 6)) - 1;
      if (!p->has[BL_P_adaptive_num_regions])
        throw ParseFailure{"Could not read input file."};  // bad_optional_access inside Read()
      if (region >= p->adaptive_num_regions) return;
      if (region < 0 || region >= BL_MAX_REGIONS)
        throw ParseFailure{"Too many adaptive regions for this build."};
      switch (s) {
        case 0: p->adaptive_region_level[region] = ToInt(val); break;
        case 1: p->adaptive_region_x_min[region] = ToDouble(val); break;
        case 2: p->adaptive_region_x_max[region] = ToDouble(val); break;
        case 3: p->adaptive_region_y_min[region] = ToDouble(val); break;
        case 4: p->adaptive_region_y_max[region] = ToDouble(val); break;
      }
      p->adaptive_region_has[region] |= 1 << s;
      return;
    }
  }
  throw ParseFailure{"Unknown key (adaptive_region_" + key + ") in input file."};
}

// RGBToXYZ (src/utils/colors.cpp:24-39): sRGB255 -> XYZ1 under D65. pow is the pinned one, like every
// other libm call of the path.
void RgbToXyz(double r, double g, double b, double *x, double *y, double *z) {
  double r1 = r / 255.0;
  double g1 = g / 255.0;
  double b1 = b / 255.0;
  double lr = r1 <= 0.040449936 ? r1 / 12.92 : bl_pow((r1 + 0.055) / 1.055, 2.4);
  double lg = g1 <= 0.040449936 ? g1 / 12.92 : bl_pow((g1 + 0.055) / 1.055, 2.4);
  double lb = b1 <= 0.040449936 ? b1 / 12.92 : bl_pow((b1 + 0.055) / 1.055, 2.4);
  *x = 0.4123955889674142 * lr + 0.3575834307637148 * lg + 0.18049264738170154 * lb;
  *y = 0.21258623078559552 * lr + 0.715170303703411 * lg + 0.0722004986433362 * lb;
  *z = 0.019297215491746938 * lr + 0.11918386458084851 * lg + 0.9504971251315798 * lb;
}

// "x,y,z" into three doubles (input_reader.cpp:468-482)
void ReadTripleValues(const std::string &val, double *x, double *y, double *z) {
  size_t pos_1 = 0, pos_2 = 0;
  *x = ToDouble(val, &pos_1);
  if (pos_1 + 1 > val.size()) throw ParseFailure{"Could not read input file."};
  std::string rest_1 = val.substr(pos_1 + 1);
  *y = ToDouble(rest_1, &pos_2);
  if (pos_1 + pos_2 + 2 > val.size()) throw ParseFailure{"Could not read input file."};
  *z = ToDouble(val.substr(pos_1 + pos_2 + 2));
  if (val[pos_1] != ',' || val[pos_1 + pos_2 + 1] != ',')
    throw ParseFailure{"Invalid triple (" + val + ") in input file."};
}

bool EndsWith(const std::string &key, const char *suffix, size_t min_size) {
  const size_t n = std::strlen(suffix);
  return key.size() >= min_size && key.compare(key.size() - n, std::string::npos, suffix) == 0;
}

// render_<...> keys without the leading "render_" (render_reader.cpp:27-224): same order of tests, same
// silent dropping of indices beyond what render_num_images / render_<i>_num_features declared.
void ReadRender(bl_params *p, const std::string &key, const std::string &val) {
  auto need_images = [&]() {
    if (!p->has[BL_P_render_num_images]) throw ParseFailure{"Could not read input file."};   // bad_optional_access
  };
  if (EndsWith(key, "_num_features", 14)) {
    need_images();
    int image = ToInt(key.substr(0, key.size() - 13)) - 1;
    if (image >= p->render_num_images) return;
    if (image < 0 || image >= BL_MAX_RENDER_IMAGES) throw ParseFailure{"Too many rendered images for this build."};
    int num_features = ToInt(val);
    if (num_features > BL_MAX_RENDER_FEATURES) throw ParseFailure{"Too many render features for this build."};
    p->render_num_features[image] = num_features;
    p->render_num_features_has[image] = 1;
    for (int f = 0; f < BL_MAX_RENDER_FEATURES; f++) p->render_has[image][f] = 0;
    return;
  }
  // <image>_<feature>_<field>
  auto indices = [&](int *image, int *feature) -> bool {
    need_images();
    size_t pos = 0;
    *image = ToInt(key, &pos) - 1;
    if (pos + 1 > key.size()) throw ParseFailure{"Could not read input file."};
    *feature = ToInt(key.substr(pos + 1)) - 1;
    if (*image >= p->render_num_images) return false;
    if (*image < 0 || *image >= BL_MAX_RENDER_IMAGES) throw ParseFailure{"Too many rendered images for this build."};
    if (!p->render_num_features_has[*image]) throw ParseFailure{"Could not read input file."};   // bad_optional_access
    if (*feature >= p->render_num_features[*image]) return false;
    if (*feature < 0) throw ParseFailure{"Could not read input file."};
    return true;
  };
  int im = 0, fe = 0;
  if (EndsWith(key, "_quantity", 12)) {
    if (!indices(&im, &fe)) return;
    static const char *const names[7] = {"rho", "n_e", "p_gas", "Theta_e", "B", "sigma", "beta_inverse"};
    for (int q = 0; q < 7; q++)
      if (val == names[q]) {
        p->render_quantity[im][fe] = q;
        p->render_has[im][fe] |= BL_RENDER_HAS_QUANTITY;
        return;
      }
    throw ParseFailure{"Invalid render quantity (" + val + ") in input file."};
  }
  if (EndsWith(key, "_type", 8)) {
    if (!indices(&im, &fe)) return;
    static const char *const names[4] = {"fill", "thresh", "rise", "fall"};
    for (int t = 0; t < 4; t++)
      if (val == names[t]) {
        p->render_type[im][fe] = t;
        p->render_has[im][fe] |= BL_RENDER_HAS_TYPE;
        return;
      }
    throw ParseFailure{"Invalid render type (" + val + ") in input file."};
  }
  struct Scalar { const char *suffix; size_t min_size; double (*field)[BL_MAX_RENDER_FEATURES]; int bit; };
  const Scalar scalars[5] = {{"_min", 7, p->render_min, BL_RENDER_HAS_MIN}, {"_max", 7, p->render_max, BL_RENDER_HAS_MAX},
                             {"_thresh", 10, p->render_thresh, BL_RENDER_HAS_THRESH},
                             {"_tau_scale", 13, p->render_tau_scale, BL_RENDER_HAS_TAU_SCALE},
                             {"_opacity", 11, p->render_opacity, BL_RENDER_HAS_OPACITY}};
  for (const Scalar &sc : scalars)
    if (EndsWith(key, sc.suffix, sc.min_size)) {
      if (!indices(&im, &fe)) return;
      sc.field[im][fe] = ToDouble(val);
      p->render_has[im][fe] |= sc.bit;
      return;
    }
  if (EndsWith(key, "_rgb", 7)) {
    if (!indices(&im, &fe)) return;
    double r, g, b;
    ReadTripleValues(val, &r, &g, &b);
    RgbToXyz(r, g, b, &p->render_x[im][fe], &p->render_y[im][fe], &p->render_z[im][fe]);
    p->render_has[im][fe] |= BL_RENDER_HAS_XYZ;
    return;
  }
  if (EndsWith(key, "_xyz", 7)) {
    if (!indices(&im, &fe)) return;
    ReadTripleValues(val, &p->render_x[im][fe], &p->render_y[im][fe], &p->render_z[im][fe]);
    p->render_has[im][fe] |= BL_RENDER_HAS_XYZ;
    return;
  }
  throw ParseFailure{"Unknown key (render_" + key + ") in input file."};
}

void SetKeyValue(bl_params *p, const std::string &key, const std::string &val) {
  // Compound keys first
  if (key == "cut_plane_origin") return ReadTriple(val, p, BL_P_cut_plane_origin_x);
  if (key == "cut_plane_normal") return ReadTriple(val, p, BL_P_cut_plane_normal_x);
  if (key.compare(0, 16, "adaptive_region_") == 0) return ReadAdaptiveRegion(p, key.substr(16), val);
  if (key.compare(0, 7, "render_") == 0 && key != "render_num_images") return ReadRender(p, key.substr(7), val);

  for (int index = 0; index < BL_P_COUNT; index++) {
    const FieldInfo &f = kFields[index];
    if (IsDerivedField(index) || key != f.name) continue;
    switch (f.kind) {
      case 'B': *FieldPtr<int32_t>(p, f) = ToBool(val) ? 1 : 0; break;
      case 'I': *FieldPtr<int32_t>(p, f) = ToInt(val); break;
      case 'E': *FieldPtr<int32_t>(p, f) = ToEnum(index, val); break;
      case 'D': *FieldPtr<double>(p, f) = ToDouble(val); break;
      case 'F': *FieldPtr<float>(p, f) = ToFloat(val); break;
      case 'G': {
        double degrees = ToDouble(val);
        *FieldPtr<double>(p, f) = degrees * kPi / 180.0;
        if (index == BL_P_camera_th) {  // ReadPole, input_reader.cpp:492-500
          p->camera_pole = (degrees == 0.0 || degrees == 180.0) ? 1 : 0;
          p->has[BL_P_camera_pole] = 1;
        }
        break;
      }
      case 'S': {
        bl_str *dst = FieldPtr<bl_str>(p, f);
        if (val.size() >= BL_STR_LEN) throw ParseFailure{"String value too long for this build."};
        std::memset(dst->s, 0, BL_STR_LEN);
        std::memcpy(dst->s, val.data(), val.size());
        break;
      }
    }
    p->has[index] = 1;
    if (index == BL_P_render_num_images) {
      if (p->render_num_images > BL_MAX_RENDER_IMAGES) throw ParseFailure{"Too many rendered images for this build."};
      for (int i = 0; i < BL_MAX_RENDER_IMAGES; i++) p->render_num_features_has[i] = 0;
    }
    if (index == BL_P_adaptive_num_regions) {
      if (p->adaptive_num_regions > BL_MAX_REGIONS)
        throw ParseFailure{"Too many adaptive regions for this build."};
      for (int r = 0; r < BL_MAX_REGIONS; r++) p->adaptive_region_has[r] = 0;
    }
    return;
  }
  throw ParseFailure{"Unknown key (" + key + ") in input file."};
}

void ParseLine(bl_params *p, std::string line) {
  std::string stripped;
  stripped.reserve(line.size());
  for (unsigned char c : line)
    if (std::isspace(c) == 0) stripped.push_back(static_cast<char>(c));
  std::string::size_type pos = stripped.find('#');
  if (pos != std::string::npos) stripped.erase(pos);
  if (stripped.empty()) return;
  pos = stripped.find('=');
  if (pos == std::string::npos) throw ParseFailure{"Invalid assignment in input file."};
  SetKeyValue(p, stripped.substr(0, pos), stripped.substr(pos + 1));
}

}  // namespace

extern "C" {

void bl_params_clear(bl_params *p) {
  if (p != nullptr) std::memset(p, 0, sizeof(bl_params));
}

size_t bl_params_sizeof(void) { return sizeof(bl_params); }

int bl_params_set_line(bl_params *p, const char *line, char *err, size_t err_len) {
  if (p == nullptr || line == nullptr) return BL_E_ARG;
  try {
    ParseLine(p, line);
  } catch (const ParseFailure &failure) {
    SetError(err, err_len, failure.message);
    return BL_E_INPUT;
  }
  return BL_OK;
}

int bl_params_read_file(bl_params *p, const char *path, int *num_runs, char *err, size_t err_len) {
  if (p == nullptr || path == nullptr) return BL_E_ARG;
  bl_params_clear(p);
  std::ifstream stream(path);
  if (!stream.is_open()) {
    SetError(err, err_len, "Could not open input file.");
    return BL_E_INPUT;
  }
  try {
    for (std::string line; std::getline(stream, line);) ParseLine(p, line);
  } catch (const ParseFailure &failure) {
    SetError(err, err_len, failure.message);
    return BL_E_INPUT;
  }
  // Count runs (input_reader.cpp:418-427); missing keys there surface as bad_optional_access,
  // which main() reports as "Could not read input file."
  int runs = 1;
  if (!p->has[BL_P_model_type]) {
    SetError(err, err_len, "Could not read input file.");
    return BL_E_MISSING;
  }
  if (p->model_type == BL_MODEL_SIMULATION) {
    if (!p->has[BL_P_simulation_multiple]) {
      SetError(err, err_len, "Could not read input file.");
      return BL_E_MISSING;
    }
    if (p->simulation_multiple) {
      bool ok = p->has[BL_P_slow_light_on] &&
                (p->slow_light_on ? p->has[BL_P_slow_num_images] != 0
                                  : (p->has[BL_P_simulation_end] && p->has[BL_P_simulation_start]));
      if (!ok) {
        SetError(err, err_len, "Could not read input file.");
        return BL_E_MISSING;
      }
      runs = p->slow_light_on ? p->slow_num_images : p->simulation_end - p->simulation_start + 1;
    }
  }
  if (num_runs != nullptr) *num_runs = runs;
  return BL_OK;
}

int bl_params_get(const bl_params *p, const char *key, double *value, int *present) {
  if (p == nullptr || key == nullptr) return BL_E_ARG;
  for (int index = 0; index < BL_P_COUNT; index++) {
    const FieldInfo &f = kFields[index];
    if (std::strcmp(key, f.name) != 0) continue;
    if (present != nullptr) *present = p->has[index];
    if (value != nullptr) {
      switch (f.kind) {
        case 'B': case 'I': case 'E': *value = *FieldPtr<int32_t>(p, f); break;
        case 'D': case 'G': *value = *FieldPtr<double>(p, f); break;
        case 'F': *value = *FieldPtr<float>(p, f); break;
        default: return BL_E_ARG;
      }
    }
    return BL_OK;
  }
  return BL_E_ARG;
}

int bl_params_get_string(const bl_params *p, const char *key, char *out, size_t out_len) {
  if (p == nullptr || key == nullptr || out == nullptr || out_len == 0) return BL_E_ARG;
  for (int index = 0; index < BL_P_COUNT; index++) {
    const FieldInfo &f = kFields[index];
    if (f.kind != 'S' || std::strcmp(key, f.name) != 0) continue;
    std::snprintf(out, out_len, "%s", FieldPtr<bl_str>(p, f)->s);
    return BL_OK;
  }
  return BL_E_ARG;
}

}  // extern "C"
