// bow_kernels.hip — ORB vocabulary (DBoW2 text format) and the per-descriptor tree descent of
// OrbVocabulary::transform (reference src/vocabulary/mod.rs:117-325) on gfx950.
//
// The vocabulary lives in HBM as a child CSR (children in file order, which is what breaks distance ties),
// node descriptors, parents, word ids and weights.  transform's data-parallel part — every descriptor walks the
// tree from the root, at each node taking the child with the smallest Hamming distance — runs with 16 lanes per
// descriptor (k = 10 children per node in ORBvoc.txt, one per lane), 4 descriptors per wave.  The HashMap
// accumulation into BowVector / FeatureVector stays with the caller (2000 entries per frame).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <cmath>

#include "orbx_internal.hpp"

struct orbx_vocabulary {
  int device = 0;
  int k = 0, l = 0, n_nodes = 0, n_words = 0;
  std::vector<uint32_t> parent;      // host copies (orbx_vocab_nodes)
  std::vector<uint8_t> is_leaf;
  std::vector<uint8_t> desc;
  std::vector<double> weight;
  // device
  int* d_child_start = nullptr;      // [n_nodes + 1]
  uint32_t* d_child = nullptr;       // [n_links]
  uint8_t* d_desc = nullptr;         // [n_nodes][32]
  uint32_t* d_parent = nullptr;      // [n_nodes], root = 0xffffffff
  uint32_t* d_word = nullptr;        // [n_nodes], word id or 0 (mod.rs:247 unwrap_or(0))
  double* d_weight = nullptr;        // [n_nodes]
};

namespace {

struct D256 { unsigned long long w[4]; };
__device__ __forceinline__ D256 ld256(const uint8_t* p) {
  const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
  D256 d;
  d.w[0] = (unsigned long long)a.x | ((unsigned long long)a.y << 32); d.w[1] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  d.w[2] = (unsigned long long)b.x | ((unsigned long long)b.y << 32); d.w[3] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
  return d;
}
__device__ __forceinline__ D256 ld256_unaligned(const uint8_t* p) {
  D256 d;
  __builtin_memcpy(&d, p, 32);
  return d;
}
__device__ __forceinline__ unsigned ham(const D256& a, const D256& b) {
  return (unsigned)(__popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) + __popcll(a.w[3] ^ b.w[3]));
}

// mod.rs:230-248 (descent), :262-275 (ancestor at the FeatureVector level)
__global__ __launch_bounds__(256) void bow_transform_kernel(const int* __restrict__ child_start, const uint32_t* __restrict__ child,
                                                            const uint8_t* __restrict__ vdesc, const uint32_t* __restrict__ parent,
                                                            const uint32_t* __restrict__ word, const double* __restrict__ weight,
                                                            const uint8_t* __restrict__ desc, int n, int levels_up,
                                                            uint32_t* __restrict__ out_word, uint32_t* __restrict__ out_leaf,
                                                            uint32_t* __restrict__ out_node, double* __restrict__ out_weight) {
  const int gl = threadIdx.x & 15;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= n) return;                                       // uniform over the 16-lane group
  const D256 d = ld256_unaligned(desc + 32 * (size_t)i);
  uint32_t node = 0;
  for (;;) {
    const int cs = child_start[node], ce = child_start[node + 1];
    if (cs == ce) break;                                    // :234 children.is_empty()
    unsigned long long best = ~0ull;                        // (distance, position in the child list): first minimum wins (:238-243)
    for (int c = cs + gl; c < ce; c += 16) {
      const unsigned dist = ham(d, ld256(vdesc + 32 * (size_t)child[c]));
      const unsigned long long key = ((unsigned long long)dist << 32) | (unsigned)(c - cs);
      best = key < best ? key : best;
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(best, off); best = o < best ? o : best; }
    node = child[cs + (int)(unsigned)(best & 0xffffffffull)];
  }
  if (gl == 0) {
    out_word[i] = word[node];
    out_leaf[i] = node;
    out_weight[i] = weight[node];
    uint32_t nd = node;
    for (int s = 0; s < levels_up; ++s) {                   // :265-272
      const uint32_t pa = parent[nd];
      if (pa == 0xffffffffu) break;
      nd = pa;
    }
    out_node[i] = nd;
  }
}

bool parse_uint(const std::string& s, unsigned long long max, unsigned long long* out) {   // Rust's str::parse::<uN>
  if (s.empty()) return false;
  size_t i = s[0] == '+' ? 1 : 0;
  if (i >= s.size()) return false;
  unsigned long long v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (unsigned)(s[i] - '0');
    if (v > max) return false;
  }
  *out = v;
  return true;
}

void vocab_free_device(orbx_vocabulary* v) {
  hipFree(v->d_child_start); hipFree(v->d_child); hipFree(v->d_desc); hipFree(v->d_parent); hipFree(v->d_word); hipFree(v->d_weight);
}

// nodes -> device tables.  parent[0] is the root's (ignored); links as load_from_text makes them (:196-198): a node is
// appended to its parent's child list only when the parent already exists, i.e. parent id < own id.
// ---- BowVector / FeatureVector accumulation (mod.rs:296-325): the two maps of OrbVocabulary::transform built on the device ----------
// One block.  Keys (word << 32 | feature index) are sorted in LDS (bitonic, padded with ~0); a run of equal words is one BowVector
// entry whose weight is the sum of its features' leaf weights IN FEATURE ORDER (what `*bow.entry(word).or_insert(0.0) += weight` does
// over i = 0..rows); the L1 norm (:316-321) is summed over the entries in ASCENDING WORD order — the reference sums a HashMap's values
// in an unspecified order, this is the order stated instead — and every weight divided by it.  The same sort on (node << 32 | index)
// gives the FeatureVector as CSR: node ids ascending, the feature indices of a node ascending (= push order, :312).
constexpr int BOWV_THREADS = 1024, BOWV_MAX = 8192;
__device__ __forceinline__ void bowv_sort(unsigned long long* key, int P) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int c = tid; c < P / 2; c += BOWV_THREADS) {
        const int i = ((c & ~(j - 1)) << 1) | (c & (j - 1)), ixj = i | j;
        const unsigned long long x = key[i], y = key[ixj];
        if ((x > y) == ((i & k) == 0)) { key[i] = y; key[ixj] = x; }
      }
    }
  __syncthreads();
}
// heads of runs of equal high words -> compact positions (block-wide exclusive count); returns the number of runs
__device__ __forceinline__ int bowv_heads(const unsigned long long* key, int n, int* pos /*[n]: run index of element i*/, int* s_w) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int run = 0;
  for (int base = 0; base < n; base += BOWV_THREADS) {
    const int i = base + tid;
    const bool head = i < n && (i == 0 || (unsigned)(key[i] >> 32) != (unsigned)(key[i - 1] >> 32));
    const unsigned long long m = __ballot(head);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int before = run;
    for (int w = 0; w < wave; ++w) before += s_w[w];
    int tot = 0;
    for (int w = 0; w < BOWV_THREADS / 64; ++w) tot += s_w[w];
    if (i < n) pos[i] = before + __popcll(m & ((1ull << lane) - 1ull)) + (head ? 0 : -1);
    run += tot;
    __syncthreads();
  }
  return run;
}
__global__ __launch_bounds__(BOWV_THREADS) void bow_vectors_kernel(const uint32_t* __restrict__ word, const uint32_t* __restrict__ node,
                                                                   const double* __restrict__ weight, int n, uint32_t* __restrict__ bow_word,
                                                                   double* __restrict__ bow_weight, uint32_t* __restrict__ fv_node,
                                                                   int* __restrict__ fv_start, int* __restrict__ fv_index,
                                                                   int* __restrict__ counts /*[2]: n_bow, n_fv*/) {
  __shared__ unsigned long long key[BOWV_MAX];
  __shared__ int s_w[BOWV_THREADS / 64];
  __shared__ double s_total;
  const int tid = threadIdx.x;
  int P = 64;
  while (P < n) P <<= 1;
  int* pos = fv_index;                                             // scratch until the FeatureVector pass overwrites it
  // ---- BowVector
  for (int i = tid; i < P; i += BOWV_THREADS) key[i] = i < n ? ((unsigned long long)word[i] << 32) | (unsigned)i : ~0ull;
  bowv_sort(key, P);
  const int n_bow = bowv_heads(key, n, pos, s_w);
  for (int i = tid; i < n; i += BOWV_THREADS) {
    const unsigned w = (unsigned)(key[i] >> 32);
    if (i == 0 || w != (unsigned)(key[i - 1] >> 32)) {              // head of a run: its features in index order
      double sum = 0.0;
      for (int t = i; t < n && (unsigned)(key[t] >> 32) == w; ++t) sum += weight[(unsigned)key[t]];
      bow_word[pos[i]] = w; bow_weight[pos[i]] = sum;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    double total = 0.0;
    for (int e = 0; e < n_bow; ++e) total += bow_weight[e];         // ascending word id
    s_total = total;
    counts[0] = n_bow;
  }
  __syncthreads();
  if (s_total > 0.0)
    for (int e = tid; e < n_bow; e += BOWV_THREADS) bow_weight[e] /= s_total;
  __syncthreads();
  // ---- FeatureVector
  for (int i = tid; i < P; i += BOWV_THREADS) key[i] = i < n ? ((unsigned long long)node[i] << 32) | (unsigned)i : ~0ull;
  bowv_sort(key, P);
  // (run index of every element -> registers first: `pos` aliases fv_index, which is written next)
  const int n_fv = bowv_heads(key, n, pos, s_w);
  __syncthreads();
  int myrun[BOWV_MAX / BOWV_THREADS];
#pragma unroll
  for (int q = 0; q < BOWV_MAX / BOWV_THREADS; ++q) { const int i = tid + q * BOWV_THREADS; myrun[q] = i < n ? pos[i] : -1; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < BOWV_MAX / BOWV_THREADS; ++q) {
    const int i = tid + q * BOWV_THREADS;
    if (i < n) {
      fv_index[i] = (int)(unsigned)key[i];
      if (i == 0 || (unsigned)(key[i] >> 32) != (unsigned)(key[i - 1] >> 32)) { fv_node[myrun[q]] = (unsigned)(key[i] >> 32); fv_start[myrun[q]] = i; }
    }
  }
  if (tid == 0) { fv_start[n_fv] = n; counts[1] = n_fv; }
}

int vocab_upload(orbx_handle* h, orbx_vocabulary* v) {
  const int n = v->n_nodes;
  std::vector<int> cs((size_t)n + 1, 0);
  for (int i = 1; i < n; ++i) if (v->parent[i] < (uint32_t)i) cs[v->parent[i] + 1]++;
  for (int i = 0; i < n; ++i) cs[i + 1] += cs[i];
  std::vector<uint32_t> ch((size_t)std::max(cs[n], 1));
  {
    std::vector<int> fill(cs.begin(), cs.end() - 1);
    for (int i = 1; i < n; ++i) if (v->parent[i] < (uint32_t)i) ch[fill[v->parent[i]]++] = (uint32_t)i;   // ascending id = file order
  }
  std::vector<uint32_t> word((size_t)n, 0u), par(v->parent);
  par[0] = 0xffffffffu;
  uint32_t wc = 0;
  for (int i = 1; i < n; ++i) if (v->is_leaf[i]) word[i] = wc++;      // :188-192
  v->n_words = (int)wc;
  ORBX_HIP(h, hipSetDevice(h->device));
  v->device = h->device;
  ORBX_HIP(h, hipMalloc(&v->d_child_start, sizeof(int) * ((size_t)n + 1)));
  ORBX_HIP(h, hipMalloc(&v->d_child, sizeof(uint32_t) * ch.size()));
  ORBX_HIP(h, hipMalloc(&v->d_desc, 32 * (size_t)n));
  ORBX_HIP(h, hipMalloc(&v->d_parent, sizeof(uint32_t) * (size_t)n));
  ORBX_HIP(h, hipMalloc(&v->d_word, sizeof(uint32_t) * (size_t)n));
  ORBX_HIP(h, hipMalloc(&v->d_weight, sizeof(double) * (size_t)n));
  ORBX_HIP(h, hipMemcpy(v->d_child_start, cs.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice));
  ORBX_HIP(h, hipMemcpy(v->d_child, ch.data(), sizeof(uint32_t) * ch.size(), hipMemcpyHostToDevice));
  ORBX_HIP(h, hipMemcpy(v->d_desc, v->desc.data(), 32 * (size_t)n, hipMemcpyHostToDevice));
  ORBX_HIP(h, hipMemcpy(v->d_parent, par.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
  ORBX_HIP(h, hipMemcpy(v->d_word, word.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
  ORBX_HIP(h, hipMemcpy(v->d_weight, v->weight.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  return ORBX_OK;
}

}  // namespace

extern "C" {

int orbx_vocab_create(orbx_handle* h, int n_nodes, const uint32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                      const double* weight, int k, int l, orbx_vocabulary** out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!out || n_nodes < 1 || (n_nodes > 1 && (!parent || !is_leaf || !desc || !weight)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_vocab_create: bad argument");
  orbx_vocabulary* v = new orbx_vocabulary();
  v->k = k; v->l = l; v->n_nodes = n_nodes;
  v->parent.assign(parent ? parent : nullptr, parent ? parent + n_nodes : nullptr);
  if (v->parent.empty()) v->parent.assign(1, 0xffffffffu);
  v->is_leaf.assign((size_t)n_nodes, 0); v->desc.assign(32 * (size_t)n_nodes, 0); v->weight.assign((size_t)n_nodes, 0.0);
  for (int i = 1; i < n_nodes; ++i) { v->is_leaf[i] = is_leaf[i] != 0; v->weight[i] = weight[i]; }
  if (n_nodes > 1) memcpy(v->desc.data() + 32, desc + 32, 32 * (size_t)(n_nodes - 1));
  v->parent[0] = 0xffffffffu;
  if (int rc = vocab_upload(h, v)) { vocab_free_device(v); delete v; return rc; }
  *out = v;
  return ORBX_OK;
}

int orbx_vocab_load_text(orbx_handle* h, const char* path, orbx_vocabulary** out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!path || !out) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_vocab_load_text: bad argument");
  std::ifstream f(path);
  if (!f) return orbx_fail(h, ORBX_ERR_INVALID, "Failed to open vocabulary file: %s", path);        // mod.rs:118-119
  std::string line;
  std::vector<std::string> parts;
  auto split = [&](const std::string& s) { parts.clear(); std::istringstream is(s); std::string t; while (is >> t) parts.push_back(t); };
  if (!std::getline(f, line)) return orbx_fail(h, ORBX_ERR_INVALID, "Empty vocabulary file");       // :124-127
  split(line);
  if (parts.size() < 2) return orbx_fail(h, ORBX_ERR_INVALID, "Invalid header format, expected: k L [scoring weighting]");
  unsigned long long k, l;
  if (!parse_uint(parts[0], ~0ull, &k)) return orbx_fail(h, ORBX_ERR_INVALID, "Invalid k value");
  if (!parse_uint(parts[1], ~0ull, &l)) return orbx_fail(h, ORBX_ERR_INVALID, "Invalid L value");
  orbx_vocabulary* v = new orbx_vocabulary();
  v->k = (int)k; v->l = (int)l;
  v->parent.push_back(0xffffffffu); v->is_leaf.push_back(0); v->desc.assign(32, 0); v->weight.push_back(0.0);   // root (:150)
  size_t line_no = 1;
  while (std::getline(f, line)) {
    ++line_no;
    split(line);
    if (parts.size() < 35) continue;                                                                  // :155-157
    unsigned long long pid, b;
    if (!parse_uint(parts[0], 0xffffffffull, &pid)) { delete v; return orbx_fail(h, ORBX_ERR_INVALID, "Invalid parent_id at line %zu", line_no); }
    uint8_t d[32];
    for (int i = 0; i < 32; ++i) {
      if (!parse_uint(parts[2 + i], 255, &b)) { delete v; return orbx_fail(h, ORBX_ERR_INVALID, "Invalid descriptor byte at line %zu", line_no); }
      d[i] = (uint8_t)b;
    }
    char* end = nullptr;
    const double w = strtod(parts[34].c_str(), &end);
    if (end == parts[34].c_str() || *end != 0) { delete v; return orbx_fail(h, ORBX_ERR_INVALID, "Invalid weight at line %zu", line_no); }
    v->parent.push_back((uint32_t)pid);
    v->is_leaf.push_back(parts[1] == "1");
    v->desc.insert(v->desc.end(), d, d + 32);
    v->weight.push_back(w);
  }
  v->n_nodes = (int)v->parent.size();
  if (int rc = vocab_upload(h, v)) { vocab_free_device(v); delete v; return rc; }
  *out = v;
  return ORBX_OK;
}

void orbx_vocab_destroy(orbx_vocabulary* v) {
  if (!v) return;
  hipSetDevice(v->device);
  vocab_free_device(v);
  delete v;
}

int orbx_vocab_info(const orbx_vocabulary* v, int* k, int* l, int* n_nodes, int* n_words) {
  if (!v) return ORBX_ERR_INVALID;
  if (k) *k = v->k;
  if (l) *l = v->l;
  if (n_nodes) *n_nodes = v->n_nodes;
  if (n_words) *n_words = v->n_words;
  return ORBX_OK;
}

int orbx_vocab_nodes(const orbx_vocabulary* v, uint32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight) {
  if (!v) return ORBX_ERR_INVALID;
  if (parent) memcpy(parent, v->parent.data(), sizeof(uint32_t) * v->parent.size());
  if (is_leaf) memcpy(is_leaf, v->is_leaf.data(), v->is_leaf.size());
  if (desc) memcpy(desc, v->desc.data(), v->desc.size());
  if (weight) memcpy(weight, v->weight.data(), sizeof(double) * v->weight.size());
  return ORBX_OK;
}

int orbx_bow_transform_device(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* d_desc, int n, int levels_up, uint32_t* d_word,
                              uint32_t* d_leaf, uint32_t* d_node, double* d_weight) {
  if (!h) return ORBX_ERR_INVALID;
  if (!v || n < 0 || levels_up < 0 || (n > 0 && (!d_desc || !d_word || !d_leaf || !d_node || !d_weight)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_bow_transform: bad argument");
  if (v->device != h->device) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_bow_transform: vocabulary lives on another device");
  if (n == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  ProfScope ps(h, "bow_transform_kernel");
  hipLaunchKernelGGL(bow_transform_kernel, dim3((n + 15) / 16), dim3(256), 0, h->stream, v->d_child_start, v->d_child, v->d_desc, v->d_parent,
                     v->d_word, v->d_weight, d_desc, n, levels_up, d_word, d_leaf, d_node, d_weight);
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int orbx_bow_transform(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* desc, int n, int levels_up, uint32_t* out_word,
                       uint32_t* out_leaf, uint32_t* out_node, double* out_weight) {
  if (!h) return ORBX_ERR_INVALID;
  if (!v || n < 0 || levels_up < 0 || (n > 0 && (!desc || !out_word || !out_leaf || !out_node || !out_weight)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_bow_transform: bad argument");
  if (n == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  if (int rc = orbx_reserve(h, h->ws_io[0], 32 * (size_t)n)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], (12 + 8) * (size_t)n)) return rc;
  uint8_t* d_desc = (uint8_t*)h->ws_io[0].p;
  double* d_w = (double*)h->ws_io[1].p;
  uint32_t* d_word = (uint32_t*)(d_w + n); uint32_t* d_leaf = d_word + n; uint32_t* d_node = d_leaf + n;
  ORBX_HIP(h, hipMemcpyAsync(d_desc, desc, 32 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  if (int rc = orbx_bow_transform_device(h, v, d_desc, n, levels_up, d_word, d_leaf, d_node, d_w)) return rc;
  ORBX_HIP(h, hipMemcpyAsync(out_word, d_word, 4 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(out_leaf, d_leaf, 4 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(out_node, d_node, 4 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(out_weight, d_w, 8 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

int orbx_bow_vectors_device(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* d_desc, int n, int levels_up, uint32_t* d_bow_word,
                            double* d_bow_weight, uint32_t* d_fv_node, int* d_fv_start, int* d_fv_index, int* d_counts) {
  if (!h) return ORBX_ERR_INVALID;
  if (!v || n < 0 || levels_up < 0 || !d_counts || (n > 0 && (!d_desc || !d_bow_word || !d_bow_weight || !d_fv_node || !d_fv_start || !d_fv_index)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_bow_vectors: bad argument");
  if (n > BOWV_MAX) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_bow_vectors: at most %d descriptors per call", BOWV_MAX);
  ORBX_HIP(h, hipSetDevice(h->device));
  if (n == 0) { ORBX_HIP(h, hipMemsetAsync(d_counts, 0, 2 * sizeof(int), h->stream)); return ORBX_OK; }
  if (int rc = orbx_reserve(h, h->ws_io[9], (12 + 8) * (size_t)n)) return rc;
  double* d_w = (double*)h->ws_io[9].p;
  uint32_t* d_word = (uint32_t*)(d_w + n); uint32_t* d_leaf = d_word + n; uint32_t* d_node = d_leaf + n;
  if (int rc = orbx_bow_transform_device(h, v, d_desc, n, levels_up, d_word, d_leaf, d_node, d_w)) return rc;
  ProfScope ps(h, "bow_vectors_kernel");
  hipLaunchKernelGGL(bow_vectors_kernel, dim3(1), dim3(BOWV_THREADS), 0, h->stream, d_word, d_node, d_w, n, d_bow_word, d_bow_weight, d_fv_node,
                     d_fv_start, d_fv_index, d_counts);
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int orbx_bow_vectors(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* desc, int n, int levels_up, uint32_t* bow_word, double* bow_weight,
                     int* n_bow, uint32_t* fv_node, int* fv_start, int* fv_index, int* n_fv) {
  if (!h) return ORBX_ERR_INVALID;
  if (!v || n < 0 || !n_bow || !n_fv || (n > 0 && (!desc || !bow_word || !bow_weight || !fv_node || !fv_start || !fv_index)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_bow_vectors: bad argument");
  *n_bow = 0; *n_fv = 0;
  if (n == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  const size_t nn = (size_t)n;
  if (int rc = orbx_reserve(h, h->ws_io[0], 32 * nn)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 8 * nn + 4 * nn + 4 * nn + 4 * (nn + 1) + 4 * nn + 64)) return rc;
  uint8_t* d_desc = (uint8_t*)h->ws_io[0].p;
  double* d_bw = (double*)h->ws_io[1].p;
  uint32_t* d_bword = (uint32_t*)(d_bw + n); uint32_t* d_fnode = d_bword + n;
  int* d_fstart = (int*)(d_fnode + n); int* d_findex = d_fstart + n + 1; int* d_counts = d_findex + n;
  ORBX_HIP(h, hipMemcpyAsync(d_desc, desc, 32 * nn, hipMemcpyHostToDevice, h->stream));
  if (int rc = orbx_bow_vectors_device(h, v, d_desc, n, levels_up, d_bword, d_bw, d_fnode, d_fstart, d_findex, d_counts)) return rc;
  int cnt[2];
  ORBX_HIP(h, hipMemcpyAsync(cnt, d_counts, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  *n_bow = cnt[0]; *n_fv = cnt[1];
  ORBX_HIP(h, hipMemcpy(bow_word, d_bword, 4 * (size_t)cnt[0], hipMemcpyDeviceToHost));
  ORBX_HIP(h, hipMemcpy(bow_weight, d_bw, 8 * (size_t)cnt[0], hipMemcpyDeviceToHost));
  ORBX_HIP(h, hipMemcpy(fv_node, d_fnode, 4 * (size_t)cnt[1], hipMemcpyDeviceToHost));
  ORBX_HIP(h, hipMemcpy(fv_start, d_fstart, 4 * ((size_t)cnt[1] + 1), hipMemcpyDeviceToHost));
  ORBX_HIP(h, hipMemcpy(fv_index, d_findex, 4 * nn, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

// OrbVocabulary::score (mod.rs:357-374): 1 - 0.5 * ||v1 - v2||_1 of two BowVectors given as (ascending word id, weight) arrays — the
// form orbx_bow_vectors returns.  Host arithmetic (two sorted lists of ~1000 entries); the terms are added in ascending word order over
// the union, where the reference walks two HashMaps in unspecified order.
int orbx_bow_score(const uint32_t* word1, const double* weight1, int n1, const uint32_t* word2, const double* weight2, int n2, double* score) {
  if (!score || n1 < 0 || n2 < 0 || (n1 > 0 && (!word1 || !weight1)) || (n2 > 0 && (!word2 || !weight2))) return ORBX_ERR_INVALID;
  for (int i = 1; i < n1; ++i) if (word1[i] <= word1[i - 1]) return ORBX_ERR_INVALID;
  for (int i = 1; i < n2; ++i) if (word2[i] <= word2[i - 1]) return ORBX_ERR_INVALID;
  double diff = 0.0;
  int a = 0, b = 0;
  while (a < n1 || b < n2) {
    if (b >= n2 || (a < n1 && word1[a] < word2[b])) { diff += std::fabs(weight1[a] - 0.0); ++a; }        // (w1 - 0).abs(), :364
    else if (a >= n1 || word2[b] < word1[a]) { diff += std::fabs(weight2[b]); ++b; }                      // :370
    else { diff += std::fabs(weight1[a] - weight2[b]); ++a; ++b; }
  }
  *score = 1.0 - 0.5 * diff;
  return ORBX_OK;
}

}  // extern "C"
