// euroc_io.hip — the input side of the per-frame path (reference src/io/euroc.rs): EuRoC `mav0` directory reader
// (cam0/cam1 data.csv, sensor.yaml), PNG decode to 8-bit grey and multi-threaded staging of stereo pairs into the
// caller's (pinned) buffer in exactly the [pair][2][h][w] layout orbx_process_stereo_batch consumes.  Host code only.
//
// PNG: the subset cv::imread(..., IMREAD_GRAYSCALE) needs for EuRoC — non-interlaced greyscale (colour type 0, with
// or without alpha), 8 or 16 bit (16 -> 8 by the high byte, as OpenCV's 8-bit load does).  Palette / RGB files are
// refused rather than converted with weights that might differ from libpng's.  Inflate is zlib's.
#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "orbx_internal.hpp"

struct orbx_euroc {
  std::string root;
  std::vector<uint64_t> ts0, ts1;
  std::vector<std::string> file0, file1;
  orbx_camera cam{};
  double k_right[4] = {0, 0, 0, 0};
  int w = 0, h = 0;
  std::string err;
};

namespace {

int euroc_fail(orbx_euroc* d, char* err, size_t cap, const std::string& msg) {
  if (d) d->err = msg;
  if (err && cap) { strncpy(err, msg.c_str(), cap - 1); err[cap - 1] = 0; }
  return ORBX_ERR_INVALID;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (unsigned char)s[a] <= ' ') ++a;
  while (b > a && (unsigned char)s[b - 1] <= ' ') --b;
  return s.substr(a, b - a);
}

// One csv record the way the `csv` crate reads it with has_headers(false), comment '#' (euroc.rs:190-194): fields
// split at ',', double quotes group (with "" as an escaped quote), lines whose first byte is '#' and empty lines skipped.
bool split_csv(const std::string& line, std::vector<std::string>& out) {
  out.clear();
  std::string cur;
  bool quoted = false;
  for (size_t i = 0; i < line.size(); ++i) {
    const char c = line[i];
    if (quoted) {
      if (c == '"') { if (i + 1 < line.size() && line[i + 1] == '"') { cur += '"'; ++i; } else quoted = false; }
      else cur += c;
    } else if (c == '"' && cur.empty()) quoted = true;
    else if (c == ',') { out.push_back(cur); cur.clear(); }
    else if (c != '\r') cur += c;
  }
  out.push_back(cur);
  return true;
}

bool parse_u64_dec(const std::string& s, uint64_t* out) {   // Rust str::parse::<u64>
  if (s.empty()) return false;
  size_t i = s[0] == '+' ? 1 : 0;
  if (i >= s.size()) return false;
  uint64_t v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    const uint64_t nv = v * 10 + (uint64_t)(s[i] - '0');
    if (nv / 10 != v) return false;
    v = nv;
  }
  *out = v;
  return true;
}

// load_image_list (euroc.rs:189-211)
int load_image_list(const std::string& path, std::vector<uint64_t>& ts, std::vector<std::string>& files, std::string& err) {
  std::ifstream f(path);
  if (!f) { err = "Failed to open " + path; return ORBX_ERR_INVALID; }
  std::string line;
  std::vector<std::string> rec;
  size_t nfields = 0;
  while (std::getline(f, line)) {
    if (line.empty() || line == "\r" || line[0] == '#') continue;
    split_csv(line, rec);
    if (nfields == 0) nfields = rec.size();
    else if (rec.size() != nfields) { err = path + ": CSV error: record with a different number of fields"; return ORBX_ERR_INVALID; }   // csv crate, flexible(false)
    if (rec.size() < 2) continue;                                         // :199-201
    uint64_t t;
    if (!parse_u64_dec(trim(rec[0]), &t)) { err = path + ": invalid digit found in string"; return ORBX_ERR_INVALID; }   // :202
    ts.push_back(t);
    files.push_back(trim(rec[1]));                                        // :203
  }
  return ORBX_OK;
}

// The numbers inside the first [...] after `key` (serde_yaml on EuRoC's sensor.yaml: `intrinsics: [..]`, `T_BS: ... data: [..]`)
bool yaml_numbers(const std::string& text, const std::string& key, const std::string& subkey, std::vector<double>& out) {
  size_t p = 0;
  for (;;) {                                   // key at the start of a line
    p = text.find(key + ":", p);
    if (p == std::string::npos) return false;
    if (p == 0 || text[p - 1] == '\n') break;
    ++p;
  }
  if (!subkey.empty()) {
    p = text.find(subkey + ":", p);
    if (p == std::string::npos) return false;
  }
  const size_t a = text.find('[', p), b = text.find(']', a == std::string::npos ? p : a);
  if (a == std::string::npos || b == std::string::npos) return false;
  std::string body = text.substr(a + 1, b - a - 1);
  for (char& c : body) if (c == ',' || c == '\n' || c == '\r') c = ' ';
  std::istringstream is(body);
  std::string tok;
  out.clear();
  while (is >> tok) {
    char* end = nullptr;
    const double v = strtod(tok.c_str(), &end);
    if (end == tok.c_str() || *end != 0) return false;
    out.push_back(v);
  }
  return true;
}

// T_c1_c0 = T_c1_b * T_b_c0 with T_BS = body<-sensor (euroc.rs:343-347): baseline = |translation|
double stereo_baseline(const std::vector<double>& T0, const std::vector<double>& T1) {
  // inverse of T0: R0^T, -R0^T t0; compose T1 * inv(T0): t = R1 (-R0^T t0) + t1
  double ti[3];
  for (int i = 0; i < 3; ++i) ti[i] = -(T0[0 * 4 + i] * T0[3] + T0[1 * 4 + i] * T0[7] + T0[2 * 4 + i] * T0[11]);
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = T1[i * 4 + 0] * ti[0] + T1[i * 4 + 1] * ti[1] + T1[i * 4 + 2] * ti[2] + T1[i * 4 + 3];
  return std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

}  // namespace

extern "C" {

// PNG -> 8-bit grey rows.  out == nullptr: only the size is returned.  Returns ORBX_ERR_INVALID for files that are not
// a non-interlaced greyscale PNG (8/16 bit, optional alpha) or are damaged; ORBX_ERR_CAPACITY when w > stride.
int orbx_png_decode_gray8(const uint8_t* file, size_t n, uint8_t* out, size_t stride, int* w_out, int* h_out) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!file || n < 8 + 25 || memcmp(file, sig, 8) != 0) return ORBX_ERR_INVALID;
  size_t p = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1;
  std::vector<uint8_t> z;
  bool have_hdr = false, end = false;
  while (p + 12 <= n && !end) {
    const uint32_t len = be32(file + p);
    const uint8_t* type = file + p + 4;
    if (p + 12 + (size_t)len > n) return ORBX_ERR_INVALID;
    const uint8_t* data = file + p + 8;
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return ORBX_ERR_INVALID;
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9];
      if (data[10] != 0 || data[11] != 0 || data[12] != 0) return ORBX_ERR_INVALID;   // compression, filter method, interlace
      if (!(ctype == 0 || ctype == 4) || !(depth == 8 || depth == 16) || w == 0 || h == 0 || w > 65535 || h > 65535) return ORBX_ERR_INVALID;
      have_hdr = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      z.insert(z.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    p += 12 + (size_t)len;
  }
  if (!have_hdr) return ORBX_ERR_INVALID;
  if (w_out) *w_out = (int)w;
  if (h_out) *h_out = (int)h;
  if (!out) return ORBX_OK;
  if ((size_t)w > stride) return ORBX_ERR_CAPACITY;
  const int bpp = (ctype == 4 ? 2 : 1) * (depth / 8);        // bytes per pixel in the filtered stream
  const size_t row = (size_t)w * bpp;
  std::vector<uint8_t> raw((row + 1) * (size_t)h);
  uLongf got = (uLongf)raw.size();
  if (uncompress(raw.data(), &got, z.data(), (uLong)z.size()) != Z_OK || got != raw.size()) return ORBX_ERR_INVALID;
  // reconstruction (PNG spec 9.2).  8-bit grey without alpha is rebuilt straight in the caller's rows (the previous
  // output row is the `prior` line); other layouts go through two scratch lines and keep the grey / high byte.
  std::vector<uint8_t> zero(row, 0), line[2];
  if (bpp != 1) { line[0].assign(row, 0); line[1].assign(row, 0); }
  const uint8_t* prev = zero.data();
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* src = raw.data() + (row + 1) * (size_t)y;
    const int ft = *src++;
    uint8_t* cur = bpp == 1 ? out + stride * (size_t)y : line[y & 1].data();
    const size_t B = (size_t)bpp;
    switch (ft) {
      case 0: memcpy(cur, src, row); break;
      case 1:
        for (size_t i = 0; i < B; ++i) cur[i] = src[i];
        for (size_t i = B; i < row; ++i) cur[i] = (uint8_t)(src[i] + cur[i - B]);
        break;
      case 2: for (size_t i = 0; i < row; ++i) cur[i] = (uint8_t)(src[i] + prev[i]); break;
      case 3:
        for (size_t i = 0; i < B; ++i) cur[i] = (uint8_t)(src[i] + (prev[i] >> 1));
        for (size_t i = B; i < row; ++i) cur[i] = (uint8_t)(src[i] + ((cur[i - B] + prev[i]) >> 1));
        break;
      case 4: {
        for (size_t i = 0; i < B; ++i) cur[i] = (uint8_t)(src[i] + prev[i]);      // a = c = 0 -> predictor b
        for (size_t i = B; i < row; ++i) {
          const int a = cur[i - B], b = prev[i], c = prev[i - B];
          const int pb0 = a - c, pa0 = b - c;                                      // p - b, p - a
          const int pa = pa0 < 0 ? -pa0 : pa0, pb = pb0 < 0 ? -pb0 : pb0, pc0 = pa0 + pb0, pc = pc0 < 0 ? -pc0 : pc0;
          const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          cur[i] = (uint8_t)(src[i] + pr);
        }
        break;
      }
      default: return ORBX_ERR_INVALID;
    }
    if (bpp != 1) {
      uint8_t* o = out + stride * (size_t)y;
      for (uint32_t x = 0; x < w; ++x) o[x] = cur[(size_t)x * bpp];   // grey (high byte of 16 bit), alpha dropped
    }
    prev = cur;
  }
  return ORBX_OK;
}

// EurocDataset::new (euroc.rs:64-90), the image side: cam0/cam1 data.csv, both sensor.yaml.  `root` is the mav0 directory.
int orbx_euroc_open(const char* root, orbx_euroc** out, char* err, size_t err_cap) {
  if (!root || !out) return ORBX_ERR_INVALID;
  orbx_euroc* d = new orbx_euroc();
  d->root = root;
  std::string e;
  if (load_image_list(d->root + "/cam0/data.csv", d->ts0, d->file0, e) || load_image_list(d->root + "/cam1/data.csv", d->ts1, d->file1, e)) {
    const int rc = euroc_fail(nullptr, err, err_cap, e);
    delete d;
    return rc;
  }
  if (d->ts0.size() != d->ts1.size()) {                                    // :69-71
    const int rc = euroc_fail(nullptr, err, err_cap, "cam0 and cam1 have different number of frames");
    delete d;
    return rc;
  }
  std::vector<double> k0, k1, T0, T1;
  for (int c = 0; c < 2; ++c) {
    const std::string yp = d->root + (c ? "/cam1/sensor.yaml" : "/cam0/sensor.yaml");
    std::vector<uint8_t> bytes;
    if (!read_file(yp, bytes)) { const int rc = euroc_fail(nullptr, err, err_cap, "Failed to open " + yp); delete d; return rc; }
    const std::string text(bytes.begin(), bytes.end());
    std::vector<double>& k = c ? k1 : k0;
    std::vector<double>& T = c ? T1 : T0;
    if (!yaml_numbers(text, "intrinsics", "", k) || !yaml_numbers(text, "T_BS", "data", T)) {
      const int rc = euroc_fail(nullptr, err, err_cap, yp + ": missing intrinsics or T_BS.data"); delete d; return rc;
    }
    if (k.size() != 4) { const int rc = euroc_fail(nullptr, err, err_cap, "Expected 4 intrinsics [fx, fy, cx, cy]"); delete d; return rc; }   // :364-369
    if (T.size() != 16) { const int rc = euroc_fail(nullptr, err, err_cap, "Expected 16 elements for transform"); delete d; return rc; }       // :379-381
  }
  d->cam = orbx_camera{k0[0], k0[1], k0[2], k0[3], stereo_baseline(T0, T1)};
  memcpy(d->k_right, k1.data(), sizeof(d->k_right));
  if (!d->file0.empty()) {
    std::vector<uint8_t> bytes;
    const std::string ip = d->root + "/cam0/data/" + d->file0[0];
    if (!read_file(ip, bytes) || orbx_png_decode_gray8(bytes.data(), bytes.size(), nullptr, 0, &d->w, &d->h) != ORBX_OK) {
      const int rc = euroc_fail(nullptr, err, err_cap, "Failed to read left image " + ip); delete d; return rc;
    }
  }
  *out = d;
  return ORBX_OK;
}

void orbx_euroc_close(orbx_euroc* d) { delete d; }
int orbx_euroc_len(const orbx_euroc* d) { return d ? (int)d->ts0.size() : 0; }                       // :92-94
const char* orbx_euroc_last_error(const orbx_euroc* d) { return d ? d->err.c_str() : ""; }

int orbx_euroc_frame_timestamp(const orbx_euroc* d, int idx, uint64_t* ts) {                         // :96-98
  if (!d || !ts || idx < 0 || (size_t)idx >= d->ts0.size()) return ORBX_ERR_INVALID;
  *ts = d->ts0[(size_t)idx];
  return ORBX_OK;
}

int orbx_euroc_calibration(const orbx_euroc* d, orbx_camera* left, double* k_right4, int* w, int* h) {
  if (!d) return ORBX_ERR_INVALID;
  if (left) *left = d->cam;
  if (k_right4) memcpy(k_right4, d->k_right, sizeof(d->k_right));
  if (w) *w = d->w;
  if (h) *h = d->h;
  return ORBX_OK;
}

// stereo_pair (euroc.rs:100-132) for `count` consecutive frames, decoded by `threads` host threads straight into
// out[pair][2][h][w] (left then right; pass a pinned buffer from orbx_host_alloc to overlap the copy to the device).
// Every image must have the size of the first one.
int orbx_euroc_read_pairs(orbx_euroc* d, int first, int count, uint8_t* out, int threads) {
  if (!d || first < 0 || count < 0 || (size_t)first + (size_t)count > d->ts0.size() || (count > 0 && !out)) {
    if (d) d->err = "orbx_euroc_read_pairs: bad range";
    return ORBX_ERR_INVALID;
  }
  const size_t img = (size_t)d->w * d->h;
  std::atomic<int> next{0}, bad{-1};
  auto work = [&]() {
    std::vector<uint8_t> bytes;
    for (;;) {
      const int j = next.fetch_add(1);
      if (j >= 2 * count) return;
      const int pr = j >> 1, side = j & 1;
      const std::string path = d->root + (side ? "/cam1/data/" + d->file1[(size_t)(first + pr)] : "/cam0/data/" + d->file0[(size_t)(first + pr)]);
      int w = 0, h = 0;
      if (!read_file(path, bytes) || orbx_png_decode_gray8(bytes.data(), bytes.size(), nullptr, 0, &w, &h) != ORBX_OK || w != d->w || h != d->h ||
          orbx_png_decode_gray8(bytes.data(), bytes.size(), out + ((size_t)pr * 2 + side) * img, (size_t)d->w, nullptr, nullptr) != ORBX_OK) {
        int expect = -1;
        bad.compare_exchange_strong(expect, j);
      }
    }
  };
  const int nt = std::max(1, std::min(threads, 2 * count));
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (bad.load() >= 0) {
    const int j = bad.load();
    d->err = std::string("Failed to read ") + ((j & 1) ? "right" : "left") + " image of frame " + std::to_string(first + (j >> 1));   // :122-125
    return ORBX_ERR_INVALID;
  }
  return ORBX_OK;
}

}  // extern "C"
