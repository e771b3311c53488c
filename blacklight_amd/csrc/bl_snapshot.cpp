// bl_snapshot.cpp - host-side snapshot reader for simulation_format = athena (Athena++ .athdf).
//
// Replaces, for that format, SimulationReader's constructor and Read() (reference
// src/simulation_reader/simulation_reader.cpp:36-159, :211-861, :870-904, :1141-1216) together with the
// subset of the HDF5 file format the reference implements by hand (hdf5_format_structure.cpp,
// hdf5_format_metadata.cpp, hdf5_format_arrays.cpp): superblock version 0 with 8-byte offsets and lengths,
// version-1 B-trees / symbol-table nodes / local heaps, version-1 object headers with continuation blocks,
// version-1 attribute, datatype and dataspace messages, version-3 contiguous data layouts. The result is
// the bl_grid_desc that bl_set_grid() takes. Error texts are the reference's where the condition is the same.
//
// Organisation (not the reference's): the file is memory-mapped once and every structure is decoded from
// bounds-checked views of the mapping; an object header is walked as a queue of message blocks, a group as a
// recursive descent through its B-tree.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <sstream>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/blacklight_amd.h"
#include "bl_internal.h"
#include "blmath.h"

namespace {

struct ReadFailure {
  int code;
  std::string message;
};

[[noreturn]] void Fail(const char *message, int code = BL_E_INPUT) { throw ReadFailure{code, message}; }

constexpr uint64_t kUndefinedAddress = ~0ull;

// ------------------------------------------------------------------------------------------------
// Bounds-checked little-endian view of the mapped file

class FileMap {
 public:
  FileMap() = default;
  FileMap(const FileMap &) = delete;
  FileMap &operator=(const FileMap &) = delete;
  ~FileMap() {
    if (base_ != nullptr && size_ > 0) munmap(const_cast<uint8_t *>(base_), size_);
  }
  void Open(const std::string &path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) Fail("Could not open file for reading.");
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
      close(fd);
      Fail("Could not open file for reading.");
    }
    size_ = static_cast<size_t>(st.st_size);
    if (size_ > 0) {
      void *p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p == MAP_FAILED) {
        close(fd);
        Fail("Could not open file for reading.");
      }
      base_ = static_cast<const uint8_t *>(p);
    }
    close(fd);
  }
  const uint8_t *At(uint64_t offset, uint64_t length) const {
    if (offset > size_ || length > size_ - offset) Fail("Unexpected end of HDF5 file.");
    return base_ + offset;
  }
  template <typename T>
  T Get(uint64_t offset) const {
    T value;
    std::memcpy(&value, At(offset, sizeof(T)), sizeof(T));
    return value;
  }
  uint8_t Byte(uint64_t offset) const { return *At(offset, 1); }
  size_t Size() const { return size_; }

 private:
  const uint8_t *base_ = nullptr;
  size_t size_ = 0;
};

// A message (or attribute payload) inside the mapping
struct Span {
  const uint8_t *data = nullptr;
  uint64_t size = 0;
  const uint8_t *At(uint64_t offset, uint64_t length) const {
    if (offset > size || length > size - offset) Fail("Unexpected end of HDF5 file.");
    return data + offset;
  }
  template <typename T>
  T Get(uint64_t offset) const {
    T value;
    std::memcpy(&value, At(offset, sizeof(T)), sizeof(T));
    return value;
  }
};

struct Message {
  uint16_t type;
  Span body;
};

// What describes a stored array: datatype message, dataspace message, element bytes
struct RawArray {
  Span datatype, dataspace, data;
};

struct GroupRef {
  uint64_t btree = kUndefinedAddress, heap = kUndefinedAddress;
};

inline uint64_t PadTo8(uint64_t n) { return (n + 7) & ~7ull; }

// ------------------------------------------------------------------------------------------------
// HDF5 structures

class Hdf5File {
 public:
  explicit Hdf5File(const std::string &path) {
    map_.Open(path);
    ReadSuperblock();
  }

  // All messages of a version-1 object header, continuation blocks followed in file order
  std::vector<Message> ObjectHeader(uint64_t address) const {
    if (map_.Byte(address) != 1) Fail("Unexpected HDF5 object header version.");
    const uint16_t num_messages = map_.Get<uint16_t>(address + 2);
    const uint32_t header_size = map_.Get<uint32_t>(address + 8);
    std::deque<std::pair<uint64_t, uint64_t>> blocks;   // (offset, length)
    blocks.emplace_back(address + 16, header_size);     // messages start on the 8-byte boundary after 12 bytes
    std::vector<Message> messages;
    messages.reserve(num_messages);
    while (!blocks.empty() && messages.size() < num_messages) {
      uint64_t pos = blocks.front().first;
      const uint64_t end = pos + blocks.front().second;
      blocks.pop_front();
      while (pos + 8 <= end && messages.size() < num_messages) {
        const uint16_t type = map_.Get<uint16_t>(pos);
        const uint16_t size = map_.Get<uint16_t>(pos + 2);
        const uint8_t flags = map_.Byte(pos + 4);
        if (flags & 0x02) Fail("Unexpected HDF5 header message flag.");   // shared message
        Message m;
        m.type = type;
        m.body.data = map_.At(pos + 8, size);
        m.body.size = size;
        if (type == 0x0010) blocks.emplace_back(m.body.Get<uint64_t>(0), m.body.Get<uint64_t>(8));
        messages.push_back(m);
        pos += 8 + static_cast<uint64_t>(size);
      }
    }
    return messages;
  }

  // Address of the object header of `path` ("a/b/c", no leading slash) below the root group
  uint64_t Locate(const std::string &path) const {
    GroupRef group = root_;
    uint64_t header = root_header_;
    size_t begin = 0;
    while (begin <= path.size()) {
      size_t slash = path.find('/', begin);
      const bool last = slash == std::string::npos;
      const std::string name = path.substr(begin, last ? std::string::npos : slash - begin);
      GroupRef child;
      if (group.btree == kUndefinedAddress || !FindInTree(group.btree, HeapData(group.heap), name, &header, &child, 0))
        Fail("Could not find HDF5 dataset in file.");
      if (last) return header;
      if (child.btree == kUndefinedAddress) child = SymbolTableOf(header);
      group = child;
      begin = slash + 1;
    }
    Fail("Could not find HDF5 dataset in file.");
  }

  RawArray Dataset(const std::string &path) const {
    RawArray raw;
    bool have_type = false, have_space = false, have_layout = false;
    for (const Message &m : ObjectHeader(Locate(path))) {
      if (m.type == 0x0003) {
        if (have_type) Fail("Too many HDF5 datatypes for dataset.");
        have_type = true;
        raw.datatype = m.body;
      } else if (m.type == 0x0001) {
        if (have_space) Fail("Too many HDF5 dataspaces for dataset.");
        have_space = true;
        raw.dataspace = m.body;
      } else if (m.type == 0x0008) {
        if (have_layout) Fail("Too many HDF5 data layouts for dataset.");
        have_layout = true;
        if (m.body.Get<uint8_t>(0) != 3) Fail("Unexpected HDF5 data layout message version.");
        if (m.body.Get<uint8_t>(1) != 1) Fail("Unexpected HDF5 data layout class.");   // contiguous only
        const uint64_t address = m.body.Get<uint64_t>(2), size = m.body.Get<uint64_t>(10);
        raw.data.size = size;
        raw.data.data = size > 0 ? map_.At(address, size) : nullptr;
      }
    }
    if (!(have_type && have_space && have_layout)) Fail("Could not find needed dataset properties.");
    return raw;
  }

  // File-level attribute by name; false if absent
  bool RootAttribute(const std::string &name, RawArray *out) const {
    for (const Message &m : ObjectHeader(root_header_)) {
      if (m.type != 0x000C) continue;
      if (m.body.Get<uint8_t>(0) != 1) Fail("Unexpected HDF5 attribute message version.");
      const uint16_t name_size = m.body.Get<uint16_t>(2);
      const uint16_t type_size = m.body.Get<uint16_t>(4);
      const uint16_t space_size = m.body.Get<uint16_t>(6);
      uint64_t offset = 8;
      const char *raw_name = reinterpret_cast<const char *>(m.body.At(offset, name_size));
      const std::string found(raw_name, name_size > 0 ? strnlen(raw_name, name_size) : 0);
      offset += PadTo8(name_size);
      if (found != name) continue;
      out->datatype.data = m.body.At(offset, type_size);
      out->datatype.size = type_size;
      offset += PadTo8(type_size);
      out->dataspace.data = m.body.At(offset, space_size);
      out->dataspace.size = space_size;
      offset += PadTo8(space_size);
      if (offset > m.body.size) Fail("Unexpected end of HDF5 file.");
      out->data.data = m.body.data + offset;
      out->data.size = m.body.size - offset;
      return true;
    }
    return false;
  }

 private:
  void ReadSuperblock() {
    static const uint8_t signature[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    if (std::memcmp(map_.At(0, 8), signature, 8) != 0) Fail("Unexpected HDF5 format signature.");
    if (map_.Byte(8) != 0) Fail("Unexpected HDF5 superblock version.");
    if (map_.Byte(9) != 0) Fail("Unexpected HDF5 file free space storage version.");
    if (map_.Byte(10) != 0) Fail("Unexpected HDF5 root group symbol table entry version.");
    if (map_.Byte(12) != 0) Fail("Unexpected HDF5 shared header message format version.");
    if (map_.Byte(13) != 8) Fail("Unexpected HDF5 size of offsets.");
    if (map_.Byte(14) != 8) Fail("Unexpected HDF5 size of lengths.");
    // 16..23 tree ranks and consistency flags, 24..55 base / free-space / end-of-file / driver addresses,
    // 56.. root group symbol table entry: name offset, header address, cache type, reserved, scratch
    root_header_ = map_.Get<uint64_t>(56 + 8);
    if (map_.Get<uint32_t>(56 + 16) != 1) Fail("Unexpected HDF5 root group symbol table entry cache type.");
    root_.btree = map_.Get<uint64_t>(56 + 24);
    root_.heap = map_.Get<uint64_t>(56 + 32);
  }

  // Address of the data segment of a local heap
  uint64_t HeapData(uint64_t heap) const {
    if (std::memcmp(map_.At(heap, 4), "HEAP", 4) != 0) Fail("Unexpected HDF5 heap signature.");
    if (map_.Byte(heap + 4) != 0) Fail("Unexpected HDF5 heap version.");
    return map_.Get<uint64_t>(heap + 24);
  }

  std::string HeapString(uint64_t heap_data, uint64_t offset) const {
    std::string s;
    for (uint64_t pos = heap_data + offset;; pos++) {
      const char c = static_cast<char>(map_.Byte(pos));
      if (c == '\0') return s;
      s.push_back(c);
    }
  }

  GroupRef SymbolTableOf(uint64_t header) const {
    for (const Message &m : ObjectHeader(header))
      if (m.type == 0x0011) {
        GroupRef g;
        g.btree = m.body.Get<uint64_t>(0);
        g.heap = m.body.Get<uint64_t>(8);
        return g;
      }
    Fail("Could not find HDF5 dataset in file.");
  }

  // Depth-first search of a version-1 group B-tree for a link name
  bool FindInTree(uint64_t node, uint64_t heap_data, const std::string &name, uint64_t *header, GroupRef *child,
                  int depth) const {
    if (depth > 64) Fail("Unexpected HDF5 B-tree signature.");
    if (std::memcmp(map_.At(node, 4), "TREE", 4) != 0) Fail("Unexpected HDF5 B-tree signature.");
    if (map_.Byte(node + 4) != 0) Fail("Unexpected HDF5 node type.");
    const uint8_t level = map_.Byte(node + 5);
    const uint16_t entries = map_.Get<uint16_t>(node + 6);
    // keys and child pointers alternate after the two sibling addresses: key 0, child 0, key 1, ...
    for (uint16_t e = 0; e < entries; e++) {
      const uint64_t child_address = map_.Get<uint64_t>(node + 24 + 8 + 16ull * e);
      if (level > 0) {
        if (FindInTree(child_address, heap_data, name, header, child, depth + 1)) return true;
        continue;
      }
      if (std::memcmp(map_.At(child_address, 4), "SNOD", 4) != 0) Fail("Unexpected HDF5 symbol table node signature.");
      if (map_.Byte(child_address + 4) != 1) Fail("Unexpected HDF5 symbol table node version.");
      const uint16_t symbols = map_.Get<uint16_t>(child_address + 6);
      for (uint16_t s = 0; s < symbols; s++) {
        const uint64_t entry = child_address + 8 + 40ull * s;
        if (HeapString(heap_data, map_.Get<uint64_t>(entry)) != name) continue;
        const uint32_t cache_type = map_.Get<uint32_t>(entry + 16);
        if (cache_type > 1) Fail("Unexpected HDF5 symbol table entry cache type.");
        *header = map_.Get<uint64_t>(entry + 8);
        *child = GroupRef{};
        if (cache_type == 1) {
          child->btree = map_.Get<uint64_t>(entry + 24);
          child->heap = map_.Get<uint64_t>(entry + 32);
        }
        return true;
      }
    }
    return false;
  }

  FileMap map_;
  uint64_t root_header_ = 0;
  GroupRef root_;
};

// ------------------------------------------------------------------------------------------------
// Array decoding

std::vector<uint64_t> Dimensions(const Span &dataspace) {
  if (dataspace.Get<uint8_t>(0) != 1) Fail("Unexpected HDF5 dataspace version.");
  const int rank = dataspace.Get<uint8_t>(1);
  if (dataspace.Get<uint8_t>(2) & 0x02) Fail("Unexpected HDF5 dataspace permutation indices.");
  std::vector<uint64_t> dims(rank);
  for (int d = 0; d < rank; d++) dims[d] = dataspace.Get<uint64_t>(8 + 8ull * d);
  return dims;
}

uint64_t Product(const std::vector<uint64_t> &dims) {
  uint64_t n = 1;
  for (uint64_t d : dims) {
    if (d != 0 && n > (1ull << 40) / d) Fail("Array dimension mismatch.");
    n *= d;
  }
  return n;
}

struct TypeHeader {
  int version, type_class;
  uint8_t bits0, bits1;
  uint32_t size;
};

TypeHeader ReadTypeHeader(const Span &datatype) {
  TypeHeader t;
  const uint8_t version_class = datatype.Get<uint8_t>(0);
  t.version = version_class >> 4;
  t.type_class = version_class & 0x0f;
  t.bits0 = datatype.Get<uint8_t>(1);
  t.bits1 = datatype.Get<uint8_t>(2);
  t.size = datatype.Get<uint32_t>(4);
  if (t.version != 1) Fail("Unexpected HDF5 datatype version.");
  return t;
}

// Element `n` of `size` bytes, byte-swapped into host order if the file order is big-endian
template <typename T>
T Element(const Span &data, uint64_t n, bool big_endian) {
  uint8_t bytes[sizeof(T)];
  std::memcpy(bytes, data.At(n * sizeof(T), sizeof(T)), sizeof(T));
  if (big_endian)
    for (size_t b = 0; b < sizeof(T) / 2; b++) std::swap(bytes[b], bytes[sizeof(T) - 1 - b]);
  T value;
  std::memcpy(&value, bytes, sizeof(T));
  return value;
}

// Fixed-point arrays of 4 or 8 bytes; 8-byte values are truncated to int32 as the reference does
std::vector<int32_t> DecodeInts(const RawArray &raw, std::vector<uint64_t> *dims_out) {
  const TypeHeader t = ReadTypeHeader(raw.datatype);
  if (t.type_class != 0) Fail("Unexpected HDF5 datatype class.");
  if (t.size != 4 && t.size != 8) Fail("Unexpected int size.");
  const bool big_endian = t.bits0 & 0x01;
  if (t.bits0 & 0x06) Fail("Unexpected HDF5 fixed-point padding.");
  if (raw.datatype.Get<uint16_t>(8) != 0 || raw.datatype.Get<uint16_t>(10) != 8 * t.size)
    Fail("Unexpected HDF5 fixed-point bit layout.");
  std::vector<uint64_t> dims = Dimensions(raw.dataspace);
  if (dims.size() > 2) Fail("Unexpected HDF5 fixed-point array size.");
  const uint64_t count = Product(dims);
  raw.data.At(0, count * t.size);   // the elements have to be there before anything is sized after them
  std::vector<int32_t> values(count);
  for (uint64_t n = 0; n < count; n++)
    values[n] = t.size == 4 ? Element<int32_t>(raw.data, n, big_endian)
                            : static_cast<int32_t>(Element<int64_t>(raw.data, n, big_endian));
  if (dims_out != nullptr) *dims_out = dims;
  return values;
}

std::vector<float> DecodeFloats(const RawArray &raw, std::vector<uint64_t> *dims_out, float *into = nullptr,
                                uint64_t expect = 0) {
  const TypeHeader t = ReadTypeHeader(raw.datatype);
  if (t.type_class != 1) Fail("Unexpected HDF5 datatype class.");
  if (t.size != 4) Fail("Unexpected float size.");
  const bool big_endian = t.bits0 & 0x01;
  if (t.bits0 & 0x40) Fail("Unexpected HDF5 floating-point byte order.");
  if (t.bits0 & 0x0e) Fail("Unexpected HDF5 floating-point padding.");
  if ((t.bits0 & 0x30) != 0x20) Fail("Unexpected HDF5 floating-point mantissa normalization.");
  const Span &d = raw.datatype;
  if (t.bits1 != 31 || d.Get<uint16_t>(8) != 0 || d.Get<uint16_t>(10) != 32 || d.Get<uint8_t>(12) != 23
      || d.Get<uint8_t>(13) != 8 || d.Get<uint8_t>(14) != 0 || d.Get<uint8_t>(15) != 23 || d.Get<uint32_t>(16) != 127)
    Fail("Unexpected HDF5 single-precision floating-point bit layout.");
  std::vector<uint64_t> dims = Dimensions(raw.dataspace);
  if (!(dims.empty() || dims.size() == 1 || dims.size() == 2 || dims.size() == 4 || dims.size() == 5))
    Fail("Unexpected HDF5 floating-point array size.");
  const uint64_t count = Product(dims);
  if (dims_out != nullptr) *dims_out = dims;
  std::vector<float> owned;
  float *dst = into;
  raw.data.At(0, count * sizeof(float));   // (as in DecodeInts)
  if (into == nullptr) {
    owned.resize(count);
    dst = owned.data();
  } else if (count != expect) {
    Fail("Array dimension mismatch.");
  }
  if (!big_endian) {
    if (count > 0) std::memcpy(dst, raw.data.At(0, count * 4), count * 4);
  } else {
    for (uint64_t n = 0; n < count; n++) dst[n] = Element<float>(raw.data, n, true);
  }
  return owned;
}

// IEEE double-precision scalars / arrays (the header values of iharm3d files)
std::vector<double> DecodeDoubles(const RawArray &raw) {
  const TypeHeader t = ReadTypeHeader(raw.datatype);
  if (t.type_class != 1) Fail("Unexpected HDF5 datatype class.");
  if (t.size != 8) Fail("Unexpected double size.");
  const bool big_endian = t.bits0 & 0x01;
  if (t.bits0 & 0x40) Fail("Unexpected HDF5 floating-point byte order.");
  if (t.bits0 & 0x0e) Fail("Unexpected HDF5 floating-point padding.");
  if ((t.bits0 & 0x30) != 0x20) Fail("Unexpected HDF5 floating-point mantissa normalization.");
  const Span &d = raw.datatype;
  if (t.bits1 != 63 || d.Get<uint16_t>(8) != 0 || d.Get<uint16_t>(10) != 64 || d.Get<uint8_t>(12) != 52
      || d.Get<uint8_t>(13) != 11 || d.Get<uint8_t>(14) != 0 || d.Get<uint8_t>(15) != 52 || d.Get<uint32_t>(16) != 1023)
    Fail("Unexpected HDF5 double-precision floating-point bit layout.");
  const uint64_t count = Product(Dimensions(raw.dataspace));
  raw.data.At(0, count * sizeof(double));   // (as in DecodeInts)
  std::vector<double> values(count);
  for (uint64_t n = 0; n < count; n++) values[n] = Element<double>(raw.data, n, big_endian);
  return values;
}

std::vector<std::string> DecodeStrings(const RawArray &raw) {
  const TypeHeader t = ReadTypeHeader(raw.datatype);
  if (t.type_class != 3) Fail("Unexpected HDF5 datatype class.");
  if ((t.bits0 >> 4) != 0) Fail("Unexpected HDF5 string encoding.");   // ASCII
  std::vector<uint64_t> dims = Dimensions(raw.dataspace);
  if (dims.size() != 1) Fail("Unexpected HDF5 string array size.");
  if (t.size == 0 || dims[0] > raw.data.size / t.size) Fail("Unexpected end of HDF5 file.");   // (sized after what is stored)
  std::vector<std::string> strings(dims[0]);
  for (uint64_t n = 0; n < dims[0]; n++) {
    const char *s = reinterpret_cast<const char *>(raw.data.At(n * t.size, t.size));
    strings[n].assign(s, strnlen(s, t.size));
  }
  return strings;
}

// ------------------------------------------------------------------------------------------------
// SimulationReader::FormatFilename (simulation_reader.cpp:870-904): "{Nd}" -> zero-padded file number

std::string FormatFilename(const std::string &pattern, int file_number) {
  const size_t open = pattern.find_first_of('{');
  if (open == std::string::npos) Fail("Invalid simulation_file for multiple runs.");
  const size_t close = pattern.find_first_of('}', open);
  if (close == std::string::npos) Fail("Invalid simulation_file for multiple runs.");
  if (pattern[close - 1] != 'd') Fail("Invalid simulation_file for multiple runs.");
  int field_length = 0;
  if (close - open > 2) {
    try {
      field_length = std::stoi(pattern.substr(open + 1, close - open - 2));
    } catch (...) {
      Fail("Invalid simulation_file for multiple runs.");
    }
  }
  const std::string number = std::to_string(file_number);
  std::string out = pattern.substr(0, open);
  if (static_cast<int>(number.size()) < field_length) out.append(field_length - number.size(), '0');
  return out + number + pattern.substr(close + 1);
}

constexpr double kPi = 3.141592653589793;               // blacklight.hpp Math::pi
constexpr double kAngularDomainTolerance = 0.1;         // simulation_reader.hpp:100

}  // namespace

// ------------------------------------------------------------------------------------------------

// The last FMKS look-up table built in this process and what it was built from (ReadIharm3d)
static std::mutex g_sks_map_mutex;
static std::vector<double> g_sks_map_key, g_sks_map;

struct bl_snapshot {
  bl_grid_desc desc{};
  std::vector<float> prim;
  std::vector<double> coords[6];   // x1f x2f x3f x1v x2v x3v
  std::vector<double> sks_map;     // simulation_coord = fmks: [2][2048][2048]
  std::vector<int32_t> levels, locations;
  double time = 0.0;
  std::string warnings, file;
  size_t setup_warning_bytes = 0;   // leading part of `warnings` that comes from the constructor's checks
};

namespace {

void Warn(bl_snapshot *s, const std::string &text) { s->warnings += "Warning: " + text + "\n"; }

constexpr const char *kMissing = "SimulationReader unable to find all needed values in input file.";

void Require(const bl_params &p, std::initializer_list<int> keys) {
  for (int k : keys)
    if (!p.has[k]) Fail(kMissing, BL_E_MISSING);
}

// Constructor part (simulation_reader.cpp:36-159): presence checks, range checks, adiabatic indices
void ReaderSetup(const bl_params &p, bl_snapshot *s) {
  Require(p, {BL_P_model_type});
  if (p.model_type != BL_MODEL_SIMULATION) Fail("Snapshots are only read in simulation mode.", BL_E_STATE);
  Require(p, {BL_P_simulation_format, BL_P_simulation_file, BL_P_simulation_multiple});
  if (p.simulation_multiple) {
    Require(p, {BL_P_simulation_start});
    if (p.simulation_start < 0) Fail("Must have nonnegative index simulation_start.");
    Require(p, {BL_P_simulation_end});
    if (p.simulation_end < p.simulation_start) Fail("Must have simulation_end at least as large as simulation_start.");
  }
  Require(p, {BL_P_simulation_coord, BL_P_simulation_a, BL_P_simulation_m_msun, BL_P_simulation_rho_cgs, BL_P_slow_light_on});
  if (p.slow_light_on) {   // :66-82
    if (!p.simulation_multiple) Fail("Must enable simulation_multiple to use slow light.");
    Require(p, {BL_P_slow_chunk_size});
    if (p.slow_chunk_size < 2) Fail("Must have slow_chunk_size be at least 2.");
    if (p.slow_chunk_size > p.simulation_end - p.simulation_start + 1) Fail("Not enough simulation files for given slow_chunk_size.");
    Require(p, {BL_P_slow_t_start, BL_P_slow_dt});
    if (p.slow_dt <= 0.0) Fail("Must have positive time interval slow_dt.");
  }
  Require(p, {BL_P_plasma_mu, BL_P_plasma_model});
  double gamma = 0.0, gamma_i = 0.0, gamma_e = 0.0;
  if (p.plasma_model == BL_PLASMA_TI_TE_BETA) {
    Require(p, {BL_P_plasma_use_p});
    if (p.plasma_use_p) {
      if (p.has[BL_P_plasma_gamma]) gamma = p.plasma_gamma;
      if (p.has[BL_P_plasma_gamma_i]) Warn(s, "Ignoring plasma_gamma_i selection.");
      if (p.has[BL_P_plasma_gamma_e]) Warn(s, "Ignoring plasma_gamma_e selection.");
    } else {
      // :108-134: athena needs all three from the input; athenak may take plasma_gamma from the file
      if (p.simulation_format == BL_SIMFMT_ATHENA) Require(p, {BL_P_plasma_gamma});
      if (p.has[BL_P_plasma_gamma]) gamma = p.plasma_gamma;
      if (p.simulation_format == BL_SIMFMT_IHARM3D) {   // may come from the file (header/gam_p, header/gam_e)
        if (p.has[BL_P_plasma_gamma_i]) gamma_i = p.plasma_gamma_i;
        if (p.has[BL_P_plasma_gamma_e]) gamma_e = p.plasma_gamma_e;
      } else {
        Require(p, {BL_P_plasma_gamma_i, BL_P_plasma_gamma_e});
        gamma_i = p.plasma_gamma_i;
        gamma_e = p.plasma_gamma_e;
      }
    }
  } else {
    Require(p, {BL_P_simulation_kappa_name});
    if (p.has[BL_P_plasma_gamma]) gamma = p.plasma_gamma;
    if (p.has[BL_P_plasma_gamma_i]) Warn(s, "Ignoring plasma_gamma_i selection.");
    if (p.has[BL_P_plasma_gamma_e]) Warn(s, "Ignoring plasma_gamma_e selection.");
  }
  s->desc.plasma_gamma = gamma;
  s->desc.plasma_gamma_i = gamma_i;
  s->desc.plasma_gamma_e = gamma_e;
}

// Variable lookup with the reference's index arithmetic (VerifyVariablesAthena, :1141-1216): indices
// count through VariableNames over all datasets, while the primitive array holds "prim" then "B".
void LocateVariables(const bl_params &p, const std::vector<std::string> &dataset_names,
                     const std::vector<std::string> &variable_names, const std::vector<int32_t> &num_variables,
                     int *ind_hydro_out, int *ind_bb_out, bl_grid_desc *d) {
  const int num_datasets = static_cast<int>(dataset_names.size());
  const int num_names = static_cast<int>(variable_names.size());
  auto find_dataset = [&](const char *name, int *offset) {
    int index = 0;
    *offset = 0;
    for (; index < num_datasets; index++) {
      if (dataset_names[index] == name) break;
      *offset += num_variables[index];
    }
    return index;
  };
  auto find_variable = [&](const std::string &name, int offset, int count) {
    int index = offset;
    for (; index < offset + count; index++)
      if (index < num_names && variable_names[index] == name) break;
    return index;
  };
  int prim_offset = 0, bb_offset = 0;
  const int ind_hydro = find_dataset("prim", &prim_offset);
  if (ind_hydro == num_datasets) Fail("Unable to locate array \"prim\" in data file.");
  const int n_hydro = num_variables[ind_hydro];
  auto hydro = [&](const std::string &name, const char *error) {
    const int index = find_variable(name, prim_offset, n_hydro);
    if (index == prim_offset + n_hydro) Fail(error);
    return index;
  };
  d->ind_rho = hydro("rho", "Unable to locate \"rho\" slice of \"prim\" in data file.");
  d->ind_pgas = hydro("press", "Unable to locate \"press\" slice of \"prim\" in data file.");
  d->ind_kappa = 0;
  if (p.plasma_model == BL_PLASMA_CODE_KAPPA)
    d->ind_kappa = hydro(p.simulation_kappa_name.s, "Unable to locate electron entropy slice of \"prim\" in data file.");
  d->ind_uu1 = hydro("vel1", "Unable to locate \"vel1\" slice of \"prim\" in data file.");
  d->ind_uu2 = hydro("vel2", "Unable to locate \"vel2\" slice of \"prim\" in data file.");
  d->ind_uu3 = hydro("vel3", "Unable to locate \"vel3\" slice of \"prim\" in data file.");
  const int ind_bb = find_dataset("B", &bb_offset);
  if (ind_bb == num_datasets) Fail("Unable to locate array \"B\" in data file.");
  const int n_bb = num_variables[ind_bb];
  auto field = [&](const std::string &name, const char *error) {
    const int index = find_variable(name, bb_offset, n_bb);
    if (index == bb_offset + n_bb) Fail(error);
    return index;
  };
  d->ind_bb1 = field("Bcc1", "Unable to locate \"Bcc1\" slice of \"prim\" in data file.");
  d->ind_bb2 = field("Bcc2", "Unable to locate \"Bcc2\" slice of \"prim\" in data file.");
  d->ind_bb3 = field("Bcc3", "Unable to locate \"Bcc3\" slice of \"prim\" in data file.");
  *ind_hydro_out = ind_hydro;
  *ind_bb_out = ind_bb;
}

std::string Scientific16(double value) {
  std::ostringstream text;
  text.setf(std::ios_base::scientific);
  text.precision(16);
  text << value;
  return text.str();
}

// file_number < 0: simulation_file as it is
void ReadAthena(const bl_params &p, int file_number, bl_snapshot *s) {
  s->file = p.simulation_file.s;
  if (file_number >= 0) s->file = FormatFilename(s->file, file_number);
  const Hdf5File file(s->file);

  // file-level attributes (hdf5_format_structure.cpp:145-289)
  RawArray root_grid_size, dataset_attr, variable_attr, count_attr, time_attr;
  if (!file.RootAttribute("RootGridSize", &root_grid_size) || !file.RootAttribute("DatasetNames", &dataset_attr)
      || !file.RootAttribute("VariableNames", &variable_attr) || !file.RootAttribute("NumVariables", &count_attr))
    Fail("Could not find needed file-level attributes.");
  const std::vector<int32_t> root_grid = DecodeInts(root_grid_size, nullptr);
  if (root_grid.size() != 3) Fail("Array dimension mismatch.");
  const std::vector<std::string> dataset_names = DecodeStrings(dataset_attr);
  const std::vector<std::string> variable_names = DecodeStrings(variable_attr);
  const std::vector<int32_t> num_variables = DecodeInts(count_attr, nullptr);
  if (num_variables.size() != dataset_names.size()) Fail("DatasetNames and NumVariables file-level attribute mismatch.");
  if (!file.RootAttribute("Time", &time_attr)) Fail("Could not find needed file-level attributes.");
  {
    std::vector<float> t = DecodeFloats(time_attr, nullptr);
    if (t.empty()) Fail("Array dimension mismatch.");
    s->time = t[0];
  }

  // block layout and coordinates (:590-620)
  std::vector<uint64_t> dims;
  s->levels = DecodeInts(file.Dataset("Levels"), &dims);
  s->locations = DecodeInts(file.Dataset("LogicalLocations"), &dims);
  const size_t n_blocks = s->levels.size();
  if (n_blocks == 0 || s->locations.size() != 3 * n_blocks) Fail("Array dimension mismatch.");
  static const char *const kCoordNames[6] = {"x1f", "x2f", "x3f", "x1v", "x2v", "x3v"};
  size_t per_block[6];
  for (int c = 0; c < 6; c++) {
    const std::vector<float> values = DecodeFloats(file.Dataset(kCoordNames[c]), &dims);
    if (dims.size() != 2 || dims[0] != n_blocks) Fail("Array dimension mismatch.");
    per_block[c] = dims[1];
    s->coords[c].assign(values.begin(), values.end());   // float32 promoted to double (:615-620)
  }
  for (int a = 0; a < 3; a++)
    if (per_block[a + 3] == 0 || per_block[a] != per_block[a + 3] + 1) Fail("Array dimension mismatch.");
  const size_t n_i = per_block[3], n_j = per_block[4], n_k = per_block[5];

  // angular ranges of single-block spherical grids (:722-758)
  if (n_blocks == 1 && p.simulation_coord == BL_COORD_SKS) {
    std::vector<double> &x2f = s->coords[1];
    const size_t last = x2f.size() - 1;
    const bool low = std::abs(x2f[0]) > (x2f[1] - x2f[0]) * kAngularDomainTolerance;
    const bool high = std::abs(x2f[last] - kPi) > (x2f[last] - x2f[last - 1]) * kAngularDomainTolerance;
    if (low || high) {
      Warn(s, "Changing theta range from [" + Scientific16(x2f[0]) + ", " + Scientific16(x2f[last]) + "] to [0, pi].");
      x2f[0] = 0.0;
      x2f[last] = kPi;
    }
  }
  if (n_blocks == 1 && (p.simulation_coord == BL_COORD_SKS || p.simulation_coord == BL_COORD_FMKS)) {
    std::vector<double> &x3f = s->coords[2];
    const size_t last = x3f.size() - 1;
    const bool low = std::abs(x3f[0]) > (x3f[1] - x3f[0]) * kAngularDomainTolerance;
    const bool high = std::abs(x3f[last] - 2.0 * kPi) > (x3f[last] - x3f[last - 1]) * kAngularDomainTolerance;
    if (low || high) {
      Warn(s, "Changing phi range from [" + Scientific16(x3f[0]) + ", " + Scientific16(x3f[last]) + "] to [0, 2*pi].");
      x3f[0] = 0.0;
      x3f[last] = 2.0 * kPi;
    }
  }

  // cell data (:761-781): "prim" variables first, then "B", whatever their order in the file
  int ind_hydro = 0, ind_bb = 0;
  LocateVariables(p, dataset_names, variable_names, num_variables, &ind_hydro, &ind_bb, &s->desc);
  const int n_hydro = num_variables[ind_hydro], n_bb = num_variables[ind_bb];
  if (n_hydro < 0 || n_bb < 0) Fail("Array dimension mismatch.");
  const size_t cells = n_blocks * n_k * n_j * n_i;
  s->prim.resize(static_cast<size_t>(n_hydro + n_bb) * cells);
  auto read_cells = [&](const char *name, int count, float *into) {
    const RawArray raw = file.Dataset(name);
    const std::vector<uint64_t> d = Dimensions(raw.dataspace);
    if (d.size() != 5 || d[0] != static_cast<uint64_t>(count) || d[1] != n_blocks || d[2] != n_k || d[3] != n_j || d[4] != n_i)
      Fail("Array dimension mismatch.");
    DecodeFloats(raw, nullptr, into, static_cast<uint64_t>(count) * cells);
  };
  read_cells("prim", n_hydro, s->prim.data());
  read_cells("B", n_bb, s->prim.data() + static_cast<size_t>(n_hydro) * cells);

  bl_grid_desc &d = s->desc;
  d.n_blocks = static_cast<int32_t>(n_blocks);
  d.n_i = static_cast<int32_t>(n_i);
  d.n_j = static_cast<int32_t>(n_j);
  d.n_k = static_cast<int32_t>(n_k);
  d.n_var = n_hydro + n_bb;
  d.prim = s->prim.data();
  d.x1f = s->coords[0].data(); d.x2f = s->coords[1].data(); d.x3f = s->coords[2].data();
  d.x1v = s->coords[3].data(); d.x2v = s->coords[4].data(); d.x3v = s->coords[5].data();
  // MeshBlock table for inter-block interpolation (hdf5_format_structure.cpp:245, simulation_reader.cpp:593-611)
  d.levels = s->levels.data();
  d.locations = s->locations.data();
  d.n_3_root = root_grid[2];
}

// ------------------------------------------------------------------------------------------------
// AthenaK binary dumps (simulation_format = athenak): ReadAthenaKHeader (simulation_reader.cpp:915-1012),
// VerifyVariablesAthenaK (:1226-1287), ReadAthenaKInputs (:1027-1131) and the block loop of Read (:434-589).
// Layout: a text header (version line, one line skipped, "  time=", one line skipped, "  size of location=",
// "  size of variable=", "  number of variables=", "  variables:", "  header offset="), `header offset` bytes of
// the run's input file, then one record per MeshBlock: six int32 cell-index bounds, logical location (3 x int32),
// level (int32), six face coordinates (float or double), then every variable over the block's cells.
// The reference counts the blocks with `while (!eof) { ignore(block); count++; }`, which ends one past the last
// block (ignore() only raises eof when it runs short), and then reads that phantom block into uninitialised arrays;
// this reader takes the blocks the file holds.
void ReadAthenaK(const bl_params &p, int file_number, bl_snapshot *s) {
  s->file = p.simulation_file.s;
  if (file_number >= 0) s->file = FormatFilename(s->file, file_number);
  FileMap map;
  map.Open(s->file);
  const size_t size = map.Size();
  const char *text = size > 0 ? reinterpret_cast<const char *>(map.At(0, size)) : "";
  size_t pos = 0;
  auto next_line = [&](bool *terminated) {   // std::istream::getline: up to, not including, the newline
    const size_t begin = pos;
    while (pos < size && text[pos] != '\n') pos++;
    std::string line(text + begin, pos - begin);
    *terminated = pos < size;
    if (pos < size) pos++;
    return line;
  };
  bool ended;
  if (next_line(&ended) != "Athena binary output version=1.1" || !ended) Fail("Unknown AthenaK file format.");
  next_line(&ended);
  auto value_after = [&](const char *label) {   // the text after `label` on the next line
    const std::string line = next_line(&ended);
    const size_t n = std::strlen(label);
    if (line.compare(0, n, label) != 0) Fail("Invalid AthenaK file header.");
    return line.substr(n);
  };
  s->time = std::strtod(value_after("  time=").c_str(), nullptr);
  next_line(&ended);
  const int location_size = std::atoi(value_after("  size of location=").c_str());
  if (location_size != 4 && location_size != 8) Fail("Unsupported size of location.");
  const int variable_size = std::atoi(value_after("  size of variable=").c_str());
  if (variable_size != 4 && variable_size != 8) Fail("Unsupported size of variables.");
  const int num_variables = std::atoi(value_after("  number of variables=").c_str());
  std::vector<std::string> names;
  {
    std::istringstream tokens(value_after("  variables:"));
    std::string name;
    while (tokens >> name) names.push_back(name);
    if (num_variables <= 0 || static_cast<int>(names.size()) < num_variables) Fail("Invalid AthenaK file header.");
    names.resize(num_variables);
  }
  const long header_offset = std::atol(value_after("  header offset=").c_str());
  if (header_offset < 0 || pos + static_cast<size_t>(header_offset) > size) Fail("Invalid AthenaK file header.");
  const size_t data_offset = pos + static_cast<size_t>(header_offset);

  // VerifyVariablesAthenaK
  const bool code_kappa = p.plasma_model == BL_PLASMA_CODE_KAPPA;
  auto locate = [&](const std::string &name, const char *message) {
    for (int n = 0; n < num_variables; n++)
      if (names[n] == name) return n;
    Fail(message);
  };
  const int in_rho = locate("dens", "Unable to locate \"dens\" values in data file.");
  const int in_pgas = locate("eint", "Unable to locate \"eint\" values in data file.");
  const int in_kappa = code_kappa ? locate(p.simulation_kappa_name.s, "Unable to locate electron entropy values in data file.") : -1;
  const int in_uu1 = locate("velx", "Unable to locate \"velx\" values in data file.");
  const int in_uu2 = locate("vely", "Unable to locate \"vely\" values in data file.");
  const int in_uu3 = locate("velz", "Unable to locate \"velz\" values in data file.");
  const int in_bb1 = locate("bcc1", "Unable to locate \"bcc1\" values in data file.");
  const int in_bb2 = locate("bcc2", "Unable to locate \"bcc2\" values in data file.");
  const int in_bb3 = locate("bcc3", "Unable to locate \"bcc3\" values in data file.");

  // ReadAthenaKInputs: the run's input file, up to the data
  const bool gamma_set = p.has[BL_P_plasma_gamma] != 0;   // every branch of the constructor that reads it sets gamma_set (:96-150)
  bool gamma_found = false;
  std::string section;
  auto mismatch = [&](const char *what, double given, double in_file) {
    std::ostringstream message;
    message << "Given " << what << " of " << given << " does not match file value of " << in_file << "; ignoring the latter.";
    Warn(s, message.str());
  };
  while (pos < data_offset) {
    const std::string line = next_line(&ended);
    if (!line.empty() && line[0] == '#') continue;
    if (!line.empty() && line.front() == '<' && line.back() == '>') {
      section = line.substr(1, line.size() - 2);
      continue;
    }
    const size_t eq = line.find('=');
    if (eq == std::string::npos) Fail("Error parsing inputs in AthenaK file.");
    std::string variable = line.substr(0, eq);
    variable.erase(std::remove(variable.begin(), variable.end(), ' '), variable.end());
    const double value = std::strtod(line.c_str() + eq + 1, nullptr);
    if (section == "coord" && variable == "a" && value != p.simulation_a) mismatch("spin", p.simulation_a, value);
    if (section == "units" && variable == "bhmass_msun" && value != p.simulation_m_msun) mismatch("mass", p.simulation_m_msun, value);
    if (section == "units" && variable == "density_cgs" && value != p.simulation_rho_cgs) mismatch("density scale", p.simulation_rho_cgs, value);
    if (section == "units" && variable == "mu" && value != p.plasma_mu) mismatch("density scale", p.plasma_mu, value);   // the reference's wording
    if (section == "mhd" && variable == "gamma") {
      if (!gamma_set)
        s->desc.plasma_gamma = value;
      else if (s->desc.plasma_gamma != value)
        mismatch("total adiabatic index", s->desc.plasma_gamma, value);
      gamma_found = true;
    }
  }
  if (!gamma_found) Fail("Missing adiabatic index.");
  pos = data_offset;

  // blocks
  const int32_t *bounds = reinterpret_cast<const int32_t *>(map.At(data_offset, 24));
  int32_t first[6];
  std::memcpy(first, bounds, 24);
  const long n_i = first[1] - first[0] + 1, n_j = first[3] - first[2] + 1, n_k = first[5] - first[4] + 1;
  if (n_i < 1 || n_j < 1 || n_k < 1 || n_i > 65536 || n_j > 65536 || n_k > 65536) Fail("Invalid AthenaK file header.");
  const size_t cells = static_cast<size_t>(n_i) * n_j * n_k;
  const size_t block_bytes = 24 + 16 + 6 * static_cast<size_t>(location_size) + static_cast<size_t>(num_variables) * cells * variable_size;
  const size_t n_blocks = (size - data_offset) / block_bytes;
  if (n_blocks == 0) Fail("Unexpected end of AthenaK file.");
  const int n_var = code_kappa ? 9 : 8;
  s->levels.assign(n_blocks, 0);
  s->locations.assign(3 * n_blocks, 0);
  const long n_axis[3] = {n_i, n_j, n_k};
  for (int a = 0; a < 3; a++) {
    s->coords[a].assign(n_blocks * (n_axis[a] + 1), 0.0);
    s->coords[3 + a].assign(n_blocks * n_axis[a], 0.0);
  }
  s->prim.assign(static_cast<size_t>(n_var) * n_blocks * cells, 0.0f);
  const int source[9] = {in_rho, in_uu1, in_uu2, in_uu3, in_pgas, in_bb1, in_bb2, in_bb3, in_kappa};   // prim order :1277-1285
  const float gamma_minus_one = static_cast<float>(s->desc.plasma_gamma - 1.0);
  for (size_t block = 0; block < n_blocks; block++) {
    const size_t begin = data_offset + block * block_bytes;
    std::memcpy(&s->locations[3 * block], map.At(begin + 24, 12), 12);
    std::memcpy(&s->levels[block], map.At(begin + 36, 4), 4);
    double faces[6];
    for (int c = 0; c < 6; c++)
      faces[c] = location_size == 4 ? static_cast<double>(map.Get<float>(begin + 40 + 4 * c)) : map.Get<double>(begin + 40 + 8 * c);
    for (int a = 0; a < 3; a++) {   // uniform faces between the block's bounds, centres midway (:508-529)
      const long n = n_axis[a];
      double *xf = &s->coords[a][block * (n + 1)], *xv = &s->coords[3 + a][block * n];
      xf[0] = faces[2 * a];
      xf[n] = faces[2 * a + 1];
      const double dx = (faces[2 * a + 1] - faces[2 * a]) / static_cast<double>(n);
      for (long i = 1; i < n; i++) xf[i] = faces[2 * a] + static_cast<double>(i) * dx;
      for (long i = 0; i < n; i++) xv[i] = 0.5 * (xf[i] + xf[i + 1]);
    }
    const size_t cell_data = begin + 40 + 6 * static_cast<size_t>(location_size);
    for (int v = 0; v < n_var; v++) {
      float *dst = &s->prim[(static_cast<size_t>(v) * n_blocks + block) * cells];
      const size_t from = cell_data + static_cast<size_t>(source[v]) * cells * variable_size;
      if (variable_size == 4) {
        std::memcpy(dst, map.At(from, cells * 4), cells * 4);
      } else {
        const uint8_t *src = map.At(from, cells * 8);
        for (size_t c = 0; c < cells; c++) {
          double value;
          std::memcpy(&value, src + 8 * c, 8);
          dst[c] = static_cast<float>(value);
        }
      }
      if (v == 4)   // internal energy to pressure (:584-589)
        for (size_t c = 0; c < cells; c++) dst[c] *= gamma_minus_one;
    }
  }
  bl_grid_desc &d = s->desc;
  d.n_blocks = static_cast<int32_t>(n_blocks);
  d.n_i = static_cast<int32_t>(n_i);
  d.n_j = static_cast<int32_t>(n_j);
  d.n_k = static_cast<int32_t>(n_k);
  d.n_var = n_var;
  d.prim = s->prim.data();
  d.x1f = s->coords[0].data(); d.x2f = s->coords[1].data(); d.x3f = s->coords[2].data();
  d.x1v = s->coords[3].data(); d.x2v = s->coords[4].data(); d.x3v = s->coords[5].data();
  d.ind_rho = 0; d.ind_uu1 = 1; d.ind_uu2 = 2; d.ind_uu3 = 3; d.ind_pgas = 4;
  d.ind_bb1 = 5; d.ind_bb2 = 6; d.ind_bb3 = 7; d.ind_kappa = code_kappa ? 8 : 0;
  d.levels = s->levels.data();
  d.locations = s->locations.data();
  d.n_3_root = 0;   // the reference sets it for HDF5 files only; it is read for spherical coordinates alone
}

// ------------------------------------------------------------------------------------------------
// iharm3d dumps (simulation_format = iharm3d) in modified Kerr-Schild coordinates (header/metric = MKS, read with
// simulation_coord = sks): Read (simulation_reader.cpp:354-428, :598-657, :782-807), VerifyVariablesHarm
// (:1302-1413), ConvertCoordinates and ConvertPrimitives3 (simulation_geometry.cpp:29-83, :95-230). One block:
// x^1 = log r, x^2 in [0, 1] with theta = pi x^2 + (1 - h) / 2 sin(2 pi x^2), x^3 = phi, all uniform;
// "prims" [n1][n2][n3][n_prim] with internal energy in place of pressure and velocity / field components on the
// modified coordinates' basis. The reader hands over a standard spherical Kerr-Schild grid: coordinates converted,
// p = (gamma - 1) u, vectors re-expressed (normal-frame velocity, lab-frame field) cell by cell.
// FMKS / MMKS dumps read with simulation_coord = fmks keep their native coordinates and get the reader's SKS -> FMKS
// look-up table and bounds instead (ConvertCoordinates, GenerateSKSMap, GetSKSCoordinates, SetJacobianFactors:
// simulation_geometry.cpp:36-57, :321-419, :427-483).
void ReadIharm3d(const bl_params &p, int file_number, bl_snapshot *s) {
  s->file = p.simulation_file.s;
  if (file_number >= 0) s->file = FormatFilename(s->file, file_number);
  const Hdf5File file(s->file);
  auto scalar = [&](const std::string &path) {
    const std::vector<double> v = DecodeDoubles(file.Dataset(path));
    if (v.empty()) Fail("Array dimension mismatch.");
    return v[0];
  };
  auto integer = [&](const std::string &path) {
    const std::vector<int32_t> v = DecodeInts(file.Dataset(path), nullptr);
    if (v.empty()) Fail("Array dimension mismatch.");
    return v[0];
  };
  s->time = scalar("t");
  // metric (:364-428)
  const std::vector<std::string> metric_names = DecodeStrings(file.Dataset("header/metric"));
  if (metric_names.empty()) Fail("Array dimension mismatch.");
  const std::string metric = metric_names[0];
  if (p.simulation_coord != BL_COORD_SKS && p.simulation_coord != BL_COORD_FMKS) Fail("Invalid simulation_coord for Harm format.");
  std::string metric_lower = metric;
  for (char &c : metric_lower) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  if (metric != "MKS" && metric != "MMKS" && metric != "FMKS")
    Warn(s, "Given metric mks does not match file value of " + metric + "; ignoring the latter.");
  const double metric_a = scalar("header/geom/" + metric_lower + "/a");
  const double metric_h = scalar("header/geom/" + metric_lower + "/hslope");
  if (metric_a != p.simulation_a) {
    std::ostringstream message;
    message << "Given spin of " << p.simulation_a << " does not match file value of " << metric_a << "; ignoring the latter.";
    Warn(s, message.str());
  }
  // FMKS / MMKS geometry (:396-427)
  const bool fmks = p.simulation_coord == BL_COORD_FMKS;
  if (fmks && metric != "MMKS" && metric != "FMKS")   // the reference would build its map from uninitialised parameters
    Fail("simulation_coord = fmks needs a file whose header/metric is FMKS or MMKS.", BL_E_UNSUPPORTED);
  double metric_r_in = 0.0, metric_poly_xt = 0.0, metric_poly_alpha = 0.0, metric_mks_smooth = 0.0, metric_derived_poly_norm = 0.0;
  if (metric == "MMKS" || metric == "FMKS") {
    bool found = true;
    try {
      metric_r_in = scalar("header/geom/" + metric_lower + "/r_in");
    } catch (const ReadFailure &) {
      found = false;
    }
    if (!found) {
      try {
        metric_r_in = scalar("header/geom/" + metric_lower + "/Rin");
      } catch (const ReadFailure &) {
        Fail("Unable to identify r_in parameter for iharm3d-format file.");
      }
    }
    metric_poly_xt = scalar("header/geom/" + metric_lower + "/poly_xt");
    metric_poly_alpha = scalar("header/geom/" + metric_lower + "/poly_alpha");
    metric_mks_smooth = scalar("header/geom/" + metric_lower + "/mks_smooth");
    metric_derived_poly_norm = (metric_poly_alpha + 1.0) * bl_pow(metric_poly_xt, metric_poly_alpha);
    metric_derived_poly_norm = 0.5 * kPi * metric_derived_poly_norm / (metric_derived_poly_norm + 1.0);
  }
  // SimulationReader::GetSKSCoordinates (simulation_geometry.cpp:427-443) and SetJacobianFactors (:451-483), FMKS branches
  auto sks_coordinates = [&](double x1, double x2, double *r, double *theta) {
    *r = bl_exp(x1);
    const double y = 2.0 * x2 - 1.0;
    const double theta_g = kPi * x2 + (1.0 - metric_h) / 2.0 * bl_sin(2.0 * kPi * x2);
    const double theta_j = 0.5 * kPi + metric_derived_poly_norm * y * (1.0 + bl_pow(y / metric_poly_xt, metric_poly_alpha) / (metric_poly_alpha + 1.0));
    *theta = theta_g + bl_exp(metric_mks_smooth * (bl_log(metric_r_in) - x1)) * (theta_j - theta_g);
  };
  auto fmks_jacobian = [&](double x1, double x2, double *dr_dx1, double *dth_dx1, double *dth_dx2) {
    *dr_dx1 = bl_exp(x1);
    const double var_a = bl_exp(metric_mks_smooth * (bl_log(metric_r_in) - x1));
    const double var_b = kPi * (0.5 - x2);
    const double var_c = bl_pow((2.0 * x2 - 1.0) / metric_poly_xt, metric_poly_alpha);
    const double var_d = 1.0 + metric_poly_alpha;
    const double var_e = metric_derived_poly_norm * (1.0 + var_c / var_d);
    const double var_f = var_e * (2.0 * x2 - 1.0);
    const double var_g = -0.5 * (1.0 - metric_h) * bl_sin(2.0 * kPi * x2);
    *dth_dx1 = -metric_mks_smooth * var_a * (var_b + var_f + var_g);
    const double var_h = kPi + (1.0 - metric_h) * kPi * bl_cos(2.0 * kPi * x2);
    const double var_i = -kPi + 2.0 * var_e;
    const double var_j = 2.0 * metric_derived_poly_norm * metric_poly_alpha * var_c / var_d;
    const double var_k = -(1.0 - metric_h) * kPi * bl_cos(2.0 * kPi * x2);
    *dth_dx2 = var_h + var_a * (var_i + var_j + var_k);
  };

  // coordinates (:622-656) and their conversion to r, theta (simulation_geometry.cpp:62-81)
  const long n[3] = {integer("header/n1"), integer("header/n2"), integer("header/n3")};
  std::vector<double> x2v_alt;
  for (int a = 0; a < 3; a++) {
    if (n[a] < 1 || n[a] > 65536) Fail("Array dimension mismatch.");
    const double start = scalar("header/geom/startx" + std::to_string(a + 1)), dx = scalar("header/geom/dx" + std::to_string(a + 1));
    std::vector<double> &xf = s->coords[a], &xv = s->coords[3 + a];
    xf.assign(n[a] + 1, 0.0);
    xv.assign(n[a], 0.0);
    xf[0] = start;
    for (long i = 0; i < n[a]; i++) {
      xf[i + 1] = start + static_cast<double>(i + 1) * dx;
      xv[i] = 0.5 * (xf[i] + xf[i + 1]);
    }
  }
  x2v_alt = s->coords[4];
  if (fmks) {
    // ConvertCoordinates, FMKS case (simulation_geometry.cpp:36-57): the coordinates stay native; the map from (r, theta)
    // to (x^1, x^2) and the grid's bounds in (r, theta, phi) are what the sampler works with. GenerateSKSMap (:321-419):
    // 2048 x 2048 points, x^2 by bisection on theta(x^1, x^2) to 1e-8.
    const int map_n1 = 2048, map_n2 = 2048, max_iter = 1000;
    const double tol = 1.0e-8;
    const double r_in = bl_exp(s->coords[0][0]), r_out = bl_exp(s->coords[0][n[0]]);
    const double dr = (r_out - r_in) / (map_n1 - 1), dtheta = kPi / (map_n2 - 1);
    // (The files of a series share their geometry - slow light opens a window of them - and the map is 4 million bisections:
    // the last map built in this process is kept with everything it was built from, and copied when those numbers come again.)
    const std::vector<double> map_key = {r_in, r_out, metric_h, metric_r_in, metric_poly_xt, metric_poly_alpha, metric_mks_smooth};
    bool map_cached = false;
    {
      std::lock_guard<std::mutex> lock(g_sks_map_mutex);
      if (g_sks_map_key == map_key && !g_sks_map.empty()) {
        s->sks_map = g_sks_map;
        map_cached = true;
      }
    }
    if (!map_cached) s->sks_map.assign(static_cast<size_t>(2) * map_n2 * map_n1, 0.0);
    double *map_x1 = s->sks_map.data(), *map_x2 = s->sks_map.data() + static_cast<size_t>(map_n2) * map_n1;
    // (columns are independent: spread over the host's threads; the reference fills the map serially)
    auto fill_columns = [&](int i_begin, int i_end) {
    for (int i = i_begin; i < i_end; ++i) {
      const double r = r_in + i * dr;
      const double x1 = bl_log(r);
      for (int j = 0; j < map_n2; ++j) {
        const double theta = std::min(j * dtheta, kPi);
        double x2 = 0.5;
        if (theta > tol && std::abs(kPi - theta) > tol) {
          double x2_a = 0.0, x2_b = 1.0;
          x2 = (x2_b + x2_a) / 2.0;
          double temp_r, theta_a = 0.0, theta_b = kPi, theta_c = kPi / 2.0;
          sks_coordinates(x1, x2_a, &temp_r, &theta_a);
          sks_coordinates(x1, x2_b, &temp_r, &theta_b);
          for (int iter = 0; iter < max_iter; iter++) {
            sks_coordinates(x1, x2, &temp_r, &theta_c);
            if ((theta_c - theta) * (theta_b - theta) < 0.0) {
              theta_a = theta_c;
              x2_a = x2;
            } else {
              theta_b = theta_c;
              x2_b = x2;
            }
            x2 = (x2_a + x2_b) / 2.0;
            if (std::abs(theta - theta_c) < tol) break;
          }
          (void)theta_a;
        } else if (theta < tol) {
          x2 = 0.0;
        } else if (theta > kPi - tol) {
          x2 = 1.0;
        }
        map_x1[static_cast<size_t>(j) * map_n1 + i] = x1;
        map_x2[static_cast<size_t>(j) * map_n1 + i] = x2;
      }
    }
    };
    if (!map_cached) {
      const int n_threads = std::max(1, std::min(64, static_cast<int>(std::thread::hardware_concurrency())));
      std::vector<std::thread> workers;
      for (int t = 0; t < n_threads; t++)
        workers.emplace_back(fill_columns, static_cast<int>(static_cast<long>(map_n1) * t / n_threads),
                             static_cast<int>(static_cast<long>(map_n1) * (t + 1) / n_threads));
      for (std::thread &w : workers) w.join();
      std::lock_guard<std::mutex> lock(g_sks_map_mutex);
      g_sks_map_key = map_key;
      g_sks_map = s->sks_map;
    }
    bl_grid_desc &gd = s->desc;
    gd.sks_map = s->sks_map.data();
    gd.sks_map_n1 = map_n1;
    gd.sks_map_n2 = map_n2;
    gd.sks_map_r_in = r_in;
    gd.sks_map_dr = dr;
    gd.sks_map_dtheta = dtheta;
    double r_val, theta_val;
    sks_coordinates(s->coords[0][0], 0.0, &r_val, &theta_val);
    gd.simulation_bounds[0] = r_val;
    gd.simulation_bounds[2] = theta_val;
    gd.simulation_bounds[4] = 0.0;
    sks_coordinates(s->coords[0][n[0]], 1.0, &r_val, &theta_val);
    gd.simulation_bounds[1] = r_val;
    gd.simulation_bounds[3] = theta_val;
    gd.simulation_bounds[5] = 2.0 * kPi;
  } else {
    for (double &x : s->coords[0]) x = bl_exp(x);
    for (double &x : s->coords[3]) x = bl_exp(x);
    for (double &x : s->coords[1]) x = kPi * x + (1.0 - metric_h) / 2.0 * bl_sin(2.0 * kPi * x);
    for (double &x : s->coords[4]) x = kPi * x + (1.0 - metric_h) / 2.0 * bl_sin(2.0 * kPi * x);
  }

  // VerifyVariablesHarm
  const int n_prim = integer("header/n_prim");
  const std::vector<std::string> names = DecodeStrings(file.Dataset("header/prim_names"));
  if (n_prim != static_cast<int>(names.size())) Fail("Inconsistency in number of primitive variables.");
  auto locate = [&](const std::string &name, const char *message) {
    for (int v = 0; v < n_prim; v++)
      if (names[v] == name) return v;
    Fail(message);
  };
  bl_grid_desc &d = s->desc;
  d.ind_rho = locate("RHO", "Unable to locate \"RHO\" slice of \"prims\" in data file.");
  d.ind_pgas = locate("UU", "Unable to locate \"UU\" slice of \"prims\" in data file.");
  d.ind_kappa = 0;
  if (p.plasma_model == BL_PLASMA_CODE_KAPPA)
    d.ind_kappa = locate(p.simulation_kappa_name.s, "Unable to locate electron entropy slice of \"prims\" in data file.");
  d.ind_uu1 = locate("U1", "Unable to locate \"U1\" slice of \"prims\" in data file.");
  d.ind_uu2 = locate("U2", "Unable to locate \"U2\" slice of \"prims\" in data file.");
  d.ind_uu3 = locate("U3", "Unable to locate \"U3\" slice of \"prims\" in data file.");
  d.ind_bb1 = locate("B1", "Unable to locate \"B1\" slice of \"prims\" in data file.");
  d.ind_bb2 = locate("B2", "Unable to locate \"B2\" slice of \"prims\" in data file.");
  d.ind_bb3 = locate("B3", "Unable to locate \"B3\" slice of \"prims\" in data file.");
  auto adiabatic_index = [&](const char *path, bool given, double *value, const char *what, const char *missing) {
    bool found = true;
    double in_file = 0.0;
    try {
      in_file = scalar(path);
    } catch (const ReadFailure &) {
      found = false;
    }
    if (!found) {
      if (!given) Fail(missing);
      return;
    }
    if (!given) {
      *value = in_file;
    } else if (*value != in_file) {
      std::ostringstream message;
      message << "Given " << what << " adiabatic index of " << *value << " does not match file value of " << in_file << "; ignoring the latter.";
      Warn(s, message.str());
    }
  };
  adiabatic_index("header/gam", p.has[BL_P_plasma_gamma] != 0, &d.plasma_gamma, "total", "Could not find total adiabatic index in input or data file.");
  if (p.plasma_model == BL_PLASMA_TI_TE_BETA && !p.plasma_use_p) {
    adiabatic_index("header/gam_p", p.has[BL_P_plasma_gamma_i] != 0, &d.plasma_gamma_i, "ion", "Could not find ion adiabatic index in input or data file.");
    adiabatic_index("header/gam_e", p.has[BL_P_plasma_gamma_e] != 0, &d.plasma_gamma_e, "electron", "Could not find electron adiabatic index in input or data file.");
  }

  // cell data: [n1][n2][n3][n_prim] -> [n_prim][1][n3][n2][n1], u -> p (:792-805)
  const size_t cells = static_cast<size_t>(n[0]) * n[1] * n[2];
  std::vector<float> transposed(cells * n_prim);
  std::vector<uint64_t> dims;
  DecodeFloats(file.Dataset("prims"), &dims, transposed.data(), transposed.size());
  if (dims.size() != 4 || dims[0] != static_cast<uint64_t>(n[0]) || dims[1] != static_cast<uint64_t>(n[1])
      || dims[2] != static_cast<uint64_t>(n[2]) || dims[3] != static_cast<uint64_t>(n_prim))
    Fail("Array dimension mismatch.");
  s->prim.assign(cells * n_prim, 0.0f);
  auto at = [&](int v, long k, long j, long i) -> float & { return s->prim[((static_cast<size_t>(v) * n[2] + k) * n[1] + j) * n[0] + i]; };
  for (int v = 0; v < n_prim; v++)
    for (long k = 0; k < n[2]; k++)
      for (long j = 0; j < n[1]; j++)
        for (long i = 0; i < n[0]; i++) at(v, k, j, i) = transposed[((static_cast<size_t>(i) * n[1] + j) * n[2] + k) * n_prim + v];
  const float gamma_minus_one = static_cast<float>(d.plasma_gamma - 1.0);
  for (size_t c = 0; c < cells; c++) s->prim[static_cast<size_t>(d.ind_pgas) * cells + c] *= gamma_minus_one;

  // ConvertPrimitives3 (simulation_geometry.cpp:95-230) for modified Kerr-Schild coordinates: the Jacobian is
  // dr/dx1 = exp(x1), dtheta/dx1 = 0, dtheta/dx2 = pi + (1 - h) pi cos(2 pi x2) (:440-468)
  const double a = p.simulation_a;
  for (long k = 0; k < n[2]; k++)
    for (long j = 0; j < n[1]; j++)
      for (long i = 0; i < n[0]; i++) {
        double r = s->coords[3][i], th = s->coords[4][j];
        double x1, x2, dr_dx1, dth_dx1, dth_dx2;
        if (fmks) {   // the cell centres are native coordinates (:118-124); Jacobian of the FMKS map (:463-475)
          x1 = r;
          x2 = th;
          sks_coordinates(x1, x2, &r, &th);
          fmks_jacobian(x1, x2, &dr_dx1, &dth_dx1, &dth_dx2);
        } else {
          x1 = bl_log(r);
          x2 = x2v_alt[j];
          dr_dx1 = bl_exp(x1);
          dth_dx1 = 0.0;
          dth_dx2 = kPi + (1.0 - metric_h) * kPi * bl_cos(2.0 * kPi * x2);
        }
        const double sth = bl_sin(th), cth = bl_cos(th);
        const double uu1 = at(d.ind_uu1, k, j, i), uu2 = at(d.ind_uu2, k, j, i), uu3 = at(d.ind_uu3, k, j, i);
        const double bb1 = at(d.ind_bb1, k, j, i), bb2 = at(d.ind_bb2, k, j, i), bb3 = at(d.ind_bb3, k, j, i);
        const double sigma = r * r + a * a * cth * cth;
        const double f = 2.0 * r / sigma;
        const double g_tr = f, g_tth = 0.0, g_tph = -a * f * sth * sth;
        const double g_rr = 1.0 + f, g_rth = 0.0, g_rph = -a * (1.0 + f) * sth * sth;
        const double g_thth = sigma, g_thph = 0.0;
        const double g_phph = (r * r + a * a + a * a * f * sth * sth) * sth * sth;
        const double gtt = -(1.0 + f), gtr = f, gtth = 0.0, gtph = 0.0;
        const double alpha = 1.0 / std::sqrt(-gtt);
        const double g_01 = dr_dx1 * g_tr + dth_dx1 * g_tth;
        const double g_02 = dth_dx2 * g_tth;
        const double g_03 = g_tph;
        const double g_11 = dr_dx1 * dr_dx1 * g_rr + 2.0 * dr_dx1 * dth_dx1 * g_rth + dth_dx1 * dth_dx1 * g_thth;
        const double g_12 = dr_dx1 * dth_dx2 * g_rth + dth_dx1 * dth_dx2 * g_thth;
        const double g_13 = dr_dx1 * g_rph + dth_dx1 * g_thph;
        const double g_22 = dth_dx2 * dth_dx2 * g_thth;
        const double g_23 = dth_dx2 * g_thph;
        const double g_33 = g_phph;
        const double g00 = gtt;
        const double g01 = gtr / dr_dx1;
        const double g02 = g_tth / dth_dx2 - dth_dx1 * g_tr / (dr_dx1 * dth_dx2);
        const double g03 = gtph;
        const double alpha_mod = 1.0 / std::sqrt(-g00);
        const double uu0 = std::sqrt(1.0 + g_11 * uu1 * uu1 + 2.0 * g_12 * uu1 * uu2 + 2.0 * g_13 * uu1 * uu3 + g_22 * uu2 * uu2
                                     + 2.0 * g_23 * uu2 * uu3 + g_33 * uu3 * uu3);
        const double u0 = uu0 / alpha_mod;
        const double u1 = uu1 - alpha_mod * g01 * uu0;
        const double u2 = uu2 - alpha_mod * g02 * uu0;
        const double u3 = uu3 - alpha_mod * g03 * uu0;
        const double u_1 = g_01 * u0 + g_11 * u1 + g_12 * u2 + g_13 * u3;
        const double u_2 = g_02 * u0 + g_12 * u1 + g_22 * u2 + g_23 * u3;
        const double u_3 = g_03 * u0 + g_13 * u1 + g_23 * u2 + g_33 * u3;
        const double ut = u0;
        const double ur = dr_dx1 * u1;
        const double uth = dth_dx1 * u1 + dth_dx2 * u2;
        const double uph = u3;
        const double uur = ur + alpha * alpha * gtr * ut;
        const double uuth = uth + alpha * alpha * gtth * ut;
        const double uuph = uph + alpha * alpha * gtph * ut;
        const double b0 = u_1 * bb1 + u_2 * bb2 + u_3 * bb3;
        const double b1 = (bb1 + b0 * u1) / u0;
        const double b2 = (bb2 + b0 * u2) / u0;
        const double b3 = (bb3 + b0 * u3) / u0;
        const double bt = b0;
        const double br = dr_dx1 * b1;
        const double bth = dth_dx1 * b1 + dth_dx2 * b2;
        const double bph = b3;
        at(d.ind_uu1, k, j, i) = static_cast<float>(uur);
        at(d.ind_uu2, k, j, i) = static_cast<float>(uuth);
        at(d.ind_uu3, k, j, i) = static_cast<float>(uuph);
        at(d.ind_bb1, k, j, i) = static_cast<float>(br * ut - bt * ur);
        at(d.ind_bb2, k, j, i) = static_cast<float>(bth * ut - bt * uth);
        at(d.ind_bb3, k, j, i) = static_cast<float>(bph * ut - bt * uph);
      }
  s->levels.assign(1, 0);
  s->locations.assign(3, 0);
  d.n_blocks = 1;
  d.n_i = static_cast<int32_t>(n[0]);
  d.n_j = static_cast<int32_t>(n[1]);
  d.n_k = static_cast<int32_t>(n[2]);
  d.n_var = n_prim;
  d.prim = s->prim.data();
  d.x1f = s->coords[0].data(); d.x2f = s->coords[1].data(); d.x3f = s->coords[2].data();
  d.x1v = s->coords[3].data(); d.x2v = s->coords[4].data(); d.x3v = s->coords[5].data();
  d.levels = s->levels.data();
  d.locations = s->locations.data();
  d.n_3_root = 0;
}

// ------------------------------------------------------------------------------------------------
// harm3d dumps (simulation_format = harm3d; Read simulation_reader.cpp:360-361, :661-716, :808-846; ConvertCoordinates
// and ConvertPrimitives4 simulation_geometry.cpp:29-83, :242-311): one line of text - time, n1 n2 n3, start and spacing
// of (log r, x2, phi), spin, adiabatic index, a radius, h-slope, a count - then float32 records of 16 (17 with an
// electron entropy) values per cell, [n1][n2][n3][...]: six coordinates, rho, u, u^mu, b^mu on the modified basis.
void ReadHarm3d(const bl_params &p, int file_number, bl_snapshot *s) {
  s->file = p.simulation_file.s;
  if (file_number >= 0) s->file = FormatFilename(s->file, file_number);
  if (p.simulation_coord != BL_COORD_SKS) Fail("Invalid simulation_coord for Harm format.");
  FileMap map;
  map.Open(s->file);
  const size_t size = map.Size();
  const char *text = size > 0 ? reinterpret_cast<const char *>(map.At(0, size)) : "";
  size_t pos = 0;
  auto number = [&]() {   // operator>> on the stream: skip white space, parse one number
    while (pos < size && std::isspace(static_cast<unsigned char>(text[pos]))) pos++;
    const size_t begin = pos;
    while (pos < size && !std::isspace(static_cast<unsigned char>(text[pos]))) pos++;
    if (begin == pos) Fail("Unexpected end of harm3d file.");
    return std::strtod(std::string(text + begin, pos - begin).c_str(), nullptr);
  };
  s->time = number();
  long n[3];
  for (int a = 0; a < 3; a++) {
    n[a] = static_cast<long>(number());
    if (n[a] < 1 || n[a] > 65536) Fail("Array dimension mismatch.");
  }
  double start[3], dx[3];
  for (int a = 0; a < 3; a++) start[a] = number();
  for (int a = 0; a < 3; a++) dx[a] = number();
  const double metric_a = number();
  if (metric_a != p.simulation_a) {
    std::ostringstream message;
    message << "Given spin of " << p.simulation_a << " does not match file value of " << metric_a << "; ignoring the latter.";
    Warn(s, message.str());
  }
  bl_grid_desc &d = s->desc;
  const double file_gamma = number();
  if (!p.has[BL_P_plasma_gamma]) {
    d.plasma_gamma = file_gamma;
  } else if (d.plasma_gamma != file_gamma) {
    std::ostringstream message;
    message << "Given total adiabatic index of " << d.plasma_gamma << " does not match file value of " << file_gamma << "; ignoring the latter.";
    Warn(s, message.str());
  }
  number();
  const double metric_h = number();
  number();
  pos += 1;   // the newline
  for (int a = 0; a < 3; a++) {
    std::vector<double> &xf = s->coords[a], &xv = s->coords[3 + a];
    xf.assign(n[a] + 1, 0.0);
    xv.assign(n[a], 0.0);
    xf[0] = start[a];
    for (long i = 0; i < n[a]; i++) {
      xf[i + 1] = start[a] + static_cast<double>(i + 1) * dx[a];
      xv[i] = 0.5 * (xf[i] + xf[i + 1]);
    }
  }
  const std::vector<double> x2v_alt = s->coords[4];
  for (double &x : s->coords[0]) x = bl_exp(x);
  for (double &x : s->coords[3]) x = bl_exp(x);
  for (double &x : s->coords[1]) x = kPi * x + (1.0 - metric_h) / 2.0 * bl_sin(2.0 * kPi * x);
  for (double &x : s->coords[4]) x = kPi * x + (1.0 - metric_h) / 2.0 * bl_sin(2.0 * kPi * x);

  const int n_var = p.plasma_model == BL_PLASMA_CODE_KAPPA ? 11 : 10;   // rho, u, u^0..3, b^0..3 (, kappa)
  const int record = n_var + 6;
  const size_t cells = static_cast<size_t>(n[0]) * n[1] * n[2];
  const uint8_t *data = map.At(pos, cells * record * 4);
  s->prim.assign(cells * n_var, 0.0f);
  auto at = [&](int v, long k, long j, long i) -> float & { return s->prim[((static_cast<size_t>(v) * n[2] + k) * n[1] + j) * n[0] + i]; };
  for (int v = 0; v < n_var; v++)
    for (long k = 0; k < n[2]; k++)
      for (long j = 0; j < n[1]; j++)
        for (long i = 0; i < n[0]; i++)
          std::memcpy(&at(v, k, j, i), data + 4 * (((static_cast<size_t>(i) * n[1] + j) * n[2] + k) * record + v + 6), 4);
  const float gamma_minus_one = static_cast<float>(d.plasma_gamma - 1.0);
  for (size_t c = 0; c < cells; c++) s->prim[cells + c] *= gamma_minus_one;
  // ConvertPrimitives4 (simulation_geometry.cpp:242-311)
  const double a = p.simulation_a;
  for (long k = 0; k < n[2]; k++)
    for (long j = 0; j < n[1]; j++)
      for (long i = 0; i < n[0]; i++) {
        const double r = s->coords[3][i], th = s->coords[4][j];
        const double cth = bl_cos(th);
        const double x2 = x2v_alt[j];
        const double u0 = at(2, k, j, i), u1 = at(3, k, j, i), u2 = at(4, k, j, i), u3 = at(5, k, j, i);
        const double b0 = at(6, k, j, i), b1 = at(7, k, j, i), b2 = at(8, k, j, i), b3 = at(9, k, j, i);
        const double dr_dx1 = r;
        const double dth_dx2 = kPi + (1.0 - metric_h) * kPi * bl_cos(2.0 * kPi * x2);
        const double sigma = r * r + a * a * cth * cth;
        const double f = 2.0 * r / sigma;
        const double gtt = -(1.0 + f), gtr = f, gtth = 0.0, gtph = 0.0;
        const double alpha = 1.0 / std::sqrt(-gtt);
        const double ut = u0, ur = dr_dx1 * u1, uth = dth_dx2 * u2, uph = u3;
        const double uur = ur + alpha * alpha * gtr * ut;
        const double uuth = uth + alpha * alpha * gtth * ut;
        const double uuph = uph + alpha * alpha * gtph * ut;
        const double bt = b0, br = dr_dx1 * b1, bth = dth_dx2 * b2, bph = b3;
        at(3, k, j, i) = static_cast<float>(uur);
        at(4, k, j, i) = static_cast<float>(uuth);
        at(5, k, j, i) = static_cast<float>(uuph);
        at(7, k, j, i) = static_cast<float>(br * ut - bt * ur);
        at(8, k, j, i) = static_cast<float>(bth * ut - bt * uth);
        at(9, k, j, i) = static_cast<float>(bph * ut - bt * uph);
      }
  s->levels.assign(1, 0);
  s->locations.assign(3, 0);
  d.n_blocks = 1;
  d.n_i = static_cast<int32_t>(n[0]);
  d.n_j = static_cast<int32_t>(n[1]);
  d.n_k = static_cast<int32_t>(n[2]);
  d.n_var = n_var;
  d.prim = s->prim.data();
  d.x1f = s->coords[0].data(); d.x2f = s->coords[1].data(); d.x3f = s->coords[2].data();
  d.x1v = s->coords[3].data(); d.x2v = s->coords[4].data(); d.x3v = s->coords[5].data();
  d.ind_rho = 0; d.ind_pgas = 1; d.ind_uu1 = 3; d.ind_uu2 = 4; d.ind_uu3 = 5; d.ind_bb1 = 7; d.ind_bb2 = 8; d.ind_bb3 = 9;
  d.ind_kappa = n_var == 11 ? 10 : 0;
  d.levels = s->levels.data();
  d.locations = s->locations.data();
  d.n_3_root = 0;
}

// Either format
void ReadSnapshot(const bl_params &p, int file_number, bl_snapshot *s) {
  if (p.simulation_format == BL_SIMFMT_ATHENAK)
    ReadAthenaK(p, file_number, s);
  else if (p.simulation_format == BL_SIMFMT_IHARM3D)
    ReadIharm3d(p, file_number, s);
  else if (p.simulation_format == BL_SIMFMT_HARM3D)
    ReadHarm3d(p, file_number, s);
  else
    ReadAthena(p, file_number, s);
}

void SetError(char *err, size_t err_len, const std::string &message) {
  if (err != nullptr && err_len > 0) std::snprintf(err, err_len, "Error: %s\n", message.c_str());
}

}  // namespace

extern "C" {

int bl_snapshot_open(const bl_params *p, int snapshot, bl_snapshot **out, char *err, size_t err_len) {
  if (p == nullptr || out == nullptr || snapshot < 0) {
    SetError(err, err_len, "bl_snapshot_open: bad argument.");
    return BL_E_ARG;
  }
  *out = nullptr;
  bl_snapshot *s = nullptr;
  try {
    s = new bl_snapshot;
    ReaderSetup(*p, s);
    s->setup_warning_bytes = s->warnings.size();
    if (p->slow_light_on) Fail("slow_light_on = true: the window of files is read by bl_slow_light_read().", BL_E_STATE);
    ReadSnapshot(*p, p->simulation_multiple ? p->simulation_start + snapshot : -1, s);   // :305-319
    *out = s;
    return BL_OK;
  } catch (const ReadFailure &f) {
    delete s;
    SetError(err, err_len, f.message);
    return f.code;
  } catch (const std::exception &e) {
    delete s;
    SetError(err, err_len, std::string("Could not read simulation file (") + e.what() + ").");
    return BL_E_INPUT;
  }
}

int bl_snapshot_open_number(const bl_params *p, int file_number, bl_snapshot **out, char *err, size_t err_len) {
  if (p == nullptr || out == nullptr || file_number < 0) {
    SetError(err, err_len, "bl_snapshot_open_number: bad argument.");
    return BL_E_ARG;
  }
  *out = nullptr;
  bl_snapshot *s = nullptr;
  try {
    s = new bl_snapshot;
    ReaderSetup(*p, s);
    s->setup_warning_bytes = s->warnings.size();
    ReadSnapshot(*p, file_number, s);
    *out = s;
    return BL_OK;
  } catch (const ReadFailure &f) {
    delete s;
    SetError(err, err_len, f.message);
    return f.code;
  } catch (const std::exception &e) {
    delete s;
    SetError(err, err_len, std::string("Could not read simulation file (") + e.what() + ").");
    return BL_E_INPUT;
  }
}

// SimulationReader::Read(snapshot) with slow light (simulation_reader.cpp:211-303, :313-861): advance the
// window of slow_chunk_size files until its latest file is not earlier than the camera time.
int bl_slow_light_read(bl_ctx *ctx, int snapshot) {
  if (ctx == nullptr || snapshot < 0) return BL_E_ARG;
  const bl_params &p = *bl_internal_params(ctx);
  bl_slow_state &state = *bl_internal_slow_state(ctx);
  constexpr double kExtrapolationTolerance = 1.0;   // simulation_reader.hpp:99
  char err[1024] = "";
  auto fail_text = [&](int code) {   // err holds "Error: ...\n"
    std::string text(err);
    if (text.rfind("Error: ", 0) == 0) text = text.substr(7);
    while (!text.empty() && text.back() == '\n') text.pop_back();
    return bl_internal_fail(ctx, code, text.c_str());
  };
  if (p.model_type != BL_MODEL_SIMULATION || !p.slow_light_on) return bl_internal_fail(ctx, BL_E_STATE, "bl_slow_light_read needs slow_light_on = true.");
  if (state.first_time) {   // the constructor's checks come before any file is touched
    try {
      bl_snapshot scratch;
      ReaderSetup(p, &scratch);
    } catch (const ReadFailure &f) {
      return bl_internal_fail(ctx, f.code, f.message.c_str());
    }
  }
  const double snapshot_time = p.slow_t_start + p.slow_dt * snapshot;
  double latest_time = snapshot_time - 2.0 * kExtrapolationTolerance;
  int latest_old = -1;
  if (state.first_time) {
    state.latest_file_number = p.simulation_start + p.slow_chunk_size - 2;
  } else {
    latest_time = state.latest_time;
    latest_old = state.latest_file_number;
  }
  // find the first file not earlier than the camera time (only its Time attribute matters here; the file is
  // read once more below if it enters the window, like the reference does)
  std::vector<bl_snapshot *> opened;   // files read while searching, by file number - latest_start
  const int search_start = state.latest_file_number + 1;
  auto close_all = [&]() {
    for (bl_snapshot *s : opened) bl_snapshot_close(s);
    opened.clear();
  };
  while (latest_time < snapshot_time && state.latest_file_number < p.simulation_end) {
    state.latest_file_number++;
    bl_snapshot *s = nullptr;
    const int rc = bl_snapshot_open_number(&p, state.latest_file_number, &s, err, sizeof err);
    if (rc != BL_OK) {
      close_all();
      return fail_text(rc);
    }
    opened.push_back(s);
    latest_time = bl_snapshot_time(s);
  }
  if (latest_time < snapshot_time - kExtrapolationTolerance) {
    close_all();
    std::ostringstream message;
    message << "Snapshot " << snapshot << " at time " << snapshot_time << " would require significant extrapolation beyond file "
            << p.simulation_end << ".";
    return bl_internal_fail(ctx, BL_E_INPUT, message.str().c_str());
  }
  if (latest_time < snapshot_time) {
    std::ostringstream message;
    message << "Snapshot " << snapshot << " at time " << snapshot_time << " requires moderate extrapolation.";
    bl_internal_warn(ctx, message.str().c_str());
  }
  // how many slices are new (:281-302)
  int num_read;
  if (state.latest_file_number == latest_old) {
    num_read = 0;
  } else if (state.latest_file_number - p.slow_chunk_size + 1 <= latest_old) {
    num_read = state.latest_file_number - latest_old;
    const int rc = bl_shift_grid_slices(ctx, num_read);
    if (rc != BL_OK) {
      close_all();
      return rc;
    }
  } else {
    num_read = p.slow_chunk_size;
  }
  // slice n <- file latest_file_number - n (:313-320)
  for (int n = 0; n < num_read; n++) {
    const int number = state.latest_file_number - n;
    bl_snapshot *s = nullptr;
    const int held = number - search_start;
    if (held >= 0 && held < static_cast<int>(opened.size())) {
      s = opened[held];
      opened[held] = nullptr;
    } else {
      const int rc = bl_snapshot_open_number(&p, number, &s, err, sizeof err);
      if (rc != BL_OK) {
        close_all();
        return fail_text(rc);
      }
    }
    if (state.first_time && n == 0) {
      const char *w = bl_snapshot_warnings(s);   // constructor and angular-range warnings, once
      std::string all(w);
      size_t at = 0;
      while (at < all.size()) {
        size_t end = all.find('\n', at);
        if (end == std::string::npos) end = all.size();
        std::string line = all.substr(at, end - at);
        if (line.rfind("Warning: ", 0) == 0) line = line.substr(9);
        if (!line.empty()) bl_internal_warn(ctx, line.c_str());
        at = end + 1;
      }
    }
    const int rc = bl_set_grid_slice(ctx, n, bl_snapshot_grid(s), bl_snapshot_time(s));
    bl_snapshot_close(s);
    if (rc != BL_OK) {
      close_all();
      return rc;
    }
  }
  close_all();
  if (num_read > 0 || state.first_time) state.latest_time = latest_time;
  state.first_time = 0;
  return bl_set_snapshot(ctx, snapshot);
}

const bl_grid_desc *bl_snapshot_grid(const bl_snapshot *s) { return s != nullptr ? &s->desc : nullptr; }

double bl_snapshot_time(const bl_snapshot *s) { return s != nullptr ? s->time : 0.0; }

const char *bl_snapshot_warnings(const bl_snapshot *s) { return s != nullptr ? s->warnings.c_str() : ""; }
size_t bl_snapshot_setup_warning_bytes(const bl_snapshot *s) { return s != nullptr ? s->setup_warning_bytes : 0; }

const char *bl_snapshot_file(const bl_snapshot *s) { return s != nullptr ? s->file.c_str() : ""; }

int bl_snapshot_blocks(const bl_snapshot *s, const int32_t **levels, const int32_t **locations) {
  if (s == nullptr) return 0;
  if (levels != nullptr) *levels = s->levels.data();
  if (locations != nullptr) *locations = s->locations.data();
  return static_cast<int>(s->levels.size());
}

void bl_snapshot_close(bl_snapshot *s) { delete s; }

}  // extern "C"
