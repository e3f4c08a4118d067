// Host side of the op-level veneer (include/q3asr_ops.h): refcounted arrays, view arithmetic, dispatch to k_ops.hip.
// One method of the reference's tch arm (src/tensor.rs:145-488, operators :960-1161) = one q3a_op_* entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/q3asr_ops.h"
#include "kernels.h"
#include "ops.h"

using namespace q3a::ops;

namespace {

thread_local std::string g_err;
[[noreturn]] void die(const std::string& m) { throw std::runtime_error(m); }
#define OHIP(expr)                                                                                   \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) die(std::string("HIP error: ") + hipGetErrorString(_e) + " (" #expr ")"); \
  } while (0)
#define OPS_TRY try {
#define OPS_CATCH                                  \
  }                                                \
  catch (const std::exception& ex) {               \
    g_err = ex.what();                             \
    return 1;                                      \
  }                                                \
  catch (...) {                                    \
    g_err = "unknown error";                       \
    return 1;                                      \
  }                                                \
  return 0;

// ---- storage: stream-ordered reuse through a per-device size-bucket cache (every op of a device runs on ONE stream) ----
std::mutex g_mu;
std::map<std::pair<int, size_t>, std::vector<void*>> g_cache;
std::map<int, hipStream_t> g_streams;

hipStream_t stream_of(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_streams.find(device);
  if (it != g_streams.end()) return it->second;
  OHIP(hipSetDevice(device));
  hipStream_t s;
  OHIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  g_streams[device] = s;
  return s;
}
size_t bucket(size_t bytes) {
  size_t b = 256;
  while (b < bytes) b <<= 1;
  return b;
}
struct Storage {
  void* p = nullptr;
  size_t cap = 0;
  int device = Q3A_CPU;
  ~Storage() {
    if (!p) return;
    if (device == Q3A_CPU) { free(p); return; }
    std::lock_guard<std::mutex> lk(g_mu);
    g_cache[{device, cap}].push_back(p);
  }
};
std::shared_ptr<Storage> alloc(size_t bytes, int device) {
  auto st = std::make_shared<Storage>();
  st->device = device;
  if (device == Q3A_CPU) {
    st->cap = std::max<size_t>(bytes, 8);
    st->p = malloc(st->cap);
    if (!st->p) die("out of host memory");
    return st;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) die("no HIP device available: libq3asr_hip has no CPU compute path");
  if (device < 0 || device >= n_dev) die("device index out of range");
  st->cap = bucket(bytes);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto& v = g_cache[{device, st->cap}];
    if (!v.empty()) { st->p = v.back(); v.pop_back(); return st; }
  }
  OHIP(hipSetDevice(device));
  OHIP(hipMalloc(&st->p, st->cap));
  return st;
}

}  // namespace

struct q3a_array {
  std::shared_ptr<Storage> st;
  int dtype = DT_F32;
  int device = Q3A_CPU;
  View v{};
  long numel() const { long n = 1; for (int d = 0; d < v.nd; ++d) n *= v.shape[d]; return n; }
  void* base() const { return st->p; }
  uint8_t* data() const { return (uint8_t*)st->p + (size_t)v.offset * dtype_size(dtype); }
};

namespace {

typedef q3a_array A;

View contig_view(const std::vector<long>& shape) {
  if ((int)shape.size() > MAXD) die("too many dimensions (max 8)");
  View v{};
  v.nd = (int)shape.size();
  long s = 1;
  for (int d = v.nd - 1; d >= 0; --d) { v.shape[d] = shape[d]; v.stride[d] = s; s *= shape[d]; }
  v.offset = 0;
  return v;
}
std::vector<long> shape_of(const A* a) { return std::vector<long>(a->v.shape, a->v.shape + a->v.nd); }
bool is_contig(const A* a) {
  long s = 1;
  for (int d = a->v.nd - 1; d >= 0; --d) {
    if (a->v.shape[d] != 1 && a->v.stride[d] != s) return false;
    s *= a->v.shape[d];
  }
  return true;
}
A* make(const std::vector<long>& shape, int dtype, int device) {
  for (long x : shape) if (x < 0) die("negative dimension");
  A* a = new A();
  a->dtype = dtype;
  a->device = device;
  a->v = contig_view(shape);
  a->st = alloc((size_t)std::max<long>(a->numel(), 1) * dtype_size(dtype), device);
  return a;
}
A* view_of(const A* a, const View& v) {
  A* r = new A(*a);
  r->v = v;
  return r;
}
int norm_dim(long dim, int nd, bool allow_end = false) {
  const long lim = allow_end ? nd + 1 : nd;
  if (dim < 0) dim += lim;
  if (dim < 0 || dim >= lim) die("dimension out of range");
  return (int)dim;
}
void need_device(const A* a, const char* op) {
  if (a->device == Q3A_CPU) die(std::string(op) + ": host array -- call to_device first (libq3asr_hip has no CPU compute path)");
}
hipStream_t sd(const A* a) { OHIP(hipSetDevice(a->device)); return stream_of(a->device); }

// host -> device on the device's own stream: a recycled block may still be read by a queued kernel of that stream, which a
// plain hipMemcpy (legacy stream) would not wait for
void h2d(void* dst, const void* src, size_t bytes, int device) {
  if (!bytes) return;
  OHIP(hipSetDevice(device));
  hipStream_t s = stream_of(device);
  OHIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
  OHIP(hipStreamSynchronize(s));
}

// copy `src` (any strides / dtype) into a fresh contiguous array of `dtype`
A* materialize(const A* src, int dtype) {
  A* out = make(shape_of(src), dtype, src->device);
  const long n = src->numel();
  if (src->device == Q3A_CPU) {
    // host arrays are small (from_slice): element loop through double
    auto ld = [&](long off) -> double {
      const uint8_t* p = (const uint8_t*)src->base();
      switch (src->dtype) {
        case DT_F32: return ((const float*)p)[off];
        case DT_I64: return (double)((const int64_t*)p)[off];
        case DT_I32: return ((const int32_t*)p)[off];
        case DT_BOOL: return p[off] ? 1.0 : 0.0;
        case DT_BF16: { uint32_t u = (uint32_t)((const uint16_t*)p)[off] << 16; float f; memcpy(&f, &u, 4); return f; }
        default: die("host conversion from this dtype is not supported");
      }
    };
    for (long i = 0; i < n; ++i) {
      long lin = i, off = src->v.offset;
      for (int d = src->v.nd - 1; d >= 0; --d) { off += (lin % src->v.shape[d]) * src->v.stride[d]; lin /= src->v.shape[d]; }
      const double x = ld(off);
      switch (dtype) {
        case DT_F32: ((float*)out->base())[i] = (float)x; break;
        case DT_I64: ((int64_t*)out->base())[i] = (int64_t)x; break;
        case DT_I32: ((int32_t*)out->base())[i] = (int32_t)x; break;
        case DT_BOOL: ((uint8_t*)out->base())[i] = x != 0.0; break;
        default: die("host conversion to this dtype is not supported");
      }
    }
    return out;
  }
  if (src->dtype == DT_C64 || dtype == DT_C64) {
    if (src->dtype != dtype) die("complex arrays convert only through abs()");
  }
  k_copy_view(out->base(), dtype, out->v, src->base(), src->dtype, src->v, n, sd(src));
  return out;
}
// F32, contiguous, on the device: what the compute kernels take
std::unique_ptr<A> f32c(const A* a, const char* op) {
  need_device(a, op);
  if (a->dtype == DT_C64) die(std::string(op) + ": complex input");
  if (a->dtype == DT_F32 && is_contig(a)) return std::unique_ptr<A>(new A(*a));
  return std::unique_ptr<A>(materialize(a, DT_F32));
}
const float* fp(const A* a) { return (const float*)a->data(); }
float* fpw(A* a) { return (float*)a->data(); }

int ret(q3a_array** out, A* r) {
  if (!out) { delete r; die("null output pointer"); }
  *out = r;
  return 0;
}

A* unary(const A* a, int op, float p, const char* name) {
  auto x = f32c(a, name);
  A* out = make(shape_of(a), DT_F32, a->device);
  k_unary(fpw(out), fp(x.get()), out->numel(), op, p, sd(a));
  return out;
}

std::vector<long> broadcast_shape(const A* a, const A* b) {
  const int nd = std::max(a->v.nd, b->v.nd);
  std::vector<long> s(nd);
  for (int i = 0; i < nd; ++i) {
    const int da = a->v.nd - nd + i, db = b->v.nd - nd + i;
    const long xa = da >= 0 ? a->v.shape[da] : 1, xb = db >= 0 ? b->v.shape[db] : 1;
    if (xa != xb && xa != 1 && xb != 1) die("shapes cannot be broadcast together");
    s[i] = xa == 1 ? xb : xa;
  }
  return s;
}
View broadcast_view(const A* a, const std::vector<long>& shape) {
  View v{};
  v.nd = (int)shape.size();
  v.offset = a->v.offset;
  for (int i = 0; i < v.nd; ++i) {
    const int da = a->v.nd - v.nd + i;
    v.shape[i] = shape[i];
    v.stride[i] = (da >= 0 && a->v.shape[da] != 1) ? a->v.stride[da] : 0;
  }
  return v;
}
A* binary(const A* a, const A* b, int op, const char* name) {
  need_device(a, name);
  need_device(b, name);
  if (a->device != b->device) die(std::string(name) + ": operands on different devices");
  std::unique_ptr<A> ka, kb;
  if (a->dtype != DT_F32) { ka.reset(materialize(a, DT_F32)); a = ka.get(); }
  if (b->dtype != DT_F32) { kb.reset(materialize(b, DT_F32)); b = kb.get(); }
  const auto shape = broadcast_shape(a, b);
  A* out = make(shape, DT_F32, a->device);
  View av = broadcast_view(a, shape), bv = broadcast_view(b, shape);
  av.offset = a->v.offset; bv.offset = b->v.offset;
  k_binary(fpw(out), (const float*)a->base(), av, (const float*)b->base(), bv, out->numel(), op, sd(a));
  return out;
}

// move `dim` to the end (view), returns the permutation's inverse through `back`
View move_dim_last(const View& v, int dim) {
  View r = v;
  int j = 0;
  for (int d = 0; d < v.nd; ++d)
    if (d != dim) { r.shape[j] = v.shape[d]; r.stride[j] = v.stride[d]; ++j; }
  r.shape[j] = v.shape[dim];
  r.stride[j] = v.stride[dim];
  return r;
}

struct DftTables { std::shared_ptr<Storage> ct, st; };
std::map<std::tuple<int, int, int>, DftTables> g_dft;

}  // namespace

extern "C" {

const char* q3a_ops_last_error(void) { return g_err.c_str(); }
int32_t q3a_ops_synchronize(int32_t device) {
  OPS_TRY
  if (device != Q3A_CPU) OHIP(hipStreamSynchronize(stream_of(device)));
  OPS_CATCH
}

void q3a_array_free(q3a_array* a) { delete a; }
int32_t q3a_op_shallow_clone(q3a_array** out, const q3a_array* a) { OPS_TRY ret(out, new A(*a)); OPS_CATCH }
int32_t q3a_array_ndim(const q3a_array* a) { return a ? a->v.nd : -1; }
int32_t q3a_array_shape(const q3a_array* a, int64_t* shape, int32_t cap) {
  if (!a) return -1;
  for (int d = 0; d < a->v.nd && d < cap; ++d) shape[d] = a->v.shape[d];
  return a->v.nd;
}
int32_t q3a_array_dtype(const q3a_array* a) { return a ? a->dtype : -1; }
int32_t q3a_array_device(const q3a_array* a) { return a ? a->device : Q3A_CPU; }
int64_t q3a_array_numel(const q3a_array* a) { return a ? a->numel() : 0; }

// ---- creation ----
int32_t q3a_op_from_slice_f32(q3a_array** out, const float* data, int64_t n) {
  OPS_TRY
  A* a = make({(long)n}, DT_F32, Q3A_CPU);
  if (n > 0) memcpy(a->base(), data, (size_t)n * 4);
  ret(out, a);
  OPS_CATCH
}
int32_t q3a_op_from_slice_i64(q3a_array** out, const int64_t* data, int64_t n) {
  OPS_TRY
  A* a = make({(long)n}, DT_I64, Q3A_CPU);
  if (n > 0) memcpy(a->base(), data, (size_t)n * 8);
  ret(out, a);
  OPS_CATCH
}
int32_t q3a_op_from_bytes(q3a_array** out, const void* data, int32_t dtype, const int64_t* shape, int32_t ndim, int32_t device) {
  OPS_TRY
  if (dtype < 0 || dtype > DT_BOOL) die("from_bytes: bad dtype");
  std::unique_ptr<A> a(make(std::vector<long>(shape, shape + ndim), dtype, device));
  const size_t bytes = (size_t)a->numel() * dtype_size(dtype);
  if (device == Q3A_CPU) memcpy(a->base(), data, bytes);
  else h2d(a->base(), data, bytes, device);
  ret(out, a.release());
  OPS_CATCH
}
static int32_t filled(q3a_array** out, const int64_t* shape, int32_t ndim, double val, int32_t dtype, int32_t device) {
  OPS_TRY
  if (dtype < 0 || dtype > DT_BOOL) die("bad dtype");
  std::unique_ptr<A> a(make(std::vector<long>(shape, shape + ndim), dtype, device));
  if (device == Q3A_CPU) {
    const long n = a->numel();
    for (long i = 0; i < n; ++i) {
      if (dtype == DT_F32) ((float*)a->base())[i] = (float)val;
      else if (dtype == DT_I64) ((int64_t*)a->base())[i] = (int64_t)val;
      else if (dtype == DT_I32) ((int32_t*)a->base())[i] = (int32_t)val;
      else if (dtype == DT_BOOL) ((uint8_t*)a->base())[i] = val != 0.0;
      else die("host fill of a 16-bit float array is not supported");
    }
  } else {
    k_fill(a->base(), dtype, a->v, a->numel(), val, sd(a.get()));
  }
  ret(out, a.release());
  OPS_CATCH
}
int32_t q3a_op_zeros(q3a_array** out, const int64_t* shape, int32_t ndim, int32_t dtype, int32_t device) { return filled(out, shape, ndim, 0.0, dtype, device); }
int32_t q3a_op_ones(q3a_array** out, const int64_t* shape, int32_t ndim, int32_t dtype, int32_t device) { return filled(out, shape, ndim, 1.0, dtype, device); }
int32_t q3a_op_full(q3a_array** out, const int64_t* shape, int32_t ndim, double val, int32_t dtype, int32_t device) { return filled(out, shape, ndim, val, dtype, device); }
int32_t q3a_op_arange(q3a_array** out, int64_t start, int64_t end, int32_t device) {
  OPS_TRY
  const long n = std::max<int64_t>(end - start, 0);
  std::unique_ptr<A> a(make({n}, DT_I64, device));
  if (device == Q3A_CPU) for (long i = 0; i < n; ++i) ((int64_t*)a->base())[i] = start + i;
  else k_arange(a->base(), DT_I64, n, (double)start, 1.0, sd(a.get()));
  ret(out, a.release());
  OPS_CATCH
}
int32_t q3a_op_arange_f(q3a_array** out, double start, double end, double step, int32_t dtype, int32_t device) {
  OPS_TRY
  if (step == 0.0) die("arange: zero step");
  const long n = std::max<long>((long)std::ceil((end - start) / step), 0);
  std::unique_ptr<A> a(make({n}, dtype, device));
  if (device == Q3A_CPU) {
    if (dtype != DT_F32) die("host arange_f: F32 only");
    for (long i = 0; i < n; ++i) ((float*)a->base())[i] = (float)(start + step * i);
  } else {
    k_arange(a->base(), dtype, n, start, step, sd(a.get()));
  }
  ret(out, a.release());
  OPS_CATCH
}
int32_t q3a_op_hann_window(q3a_array** out, int64_t size, int32_t device) {
  OPS_TRY
  std::vector<float> w((size_t)size);
  for (int64_t i = 0; i < size; ++i) w[(size_t)i] = (float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * (double)i / (double)size));  // periodic
  int64_t shp = size;
  q3a_array* r = nullptr;
  if (q3a_op_from_bytes(&r, w.data(), DT_F32, &shp, 1, device) != 0) die(g_err);
  ret(out, r);
  OPS_CATCH
}

// ---- views ----
int32_t q3a_op_contiguous(q3a_array** out, const q3a_array* a, q3a_stream*) {
  OPS_TRY
  if (is_contig(a)) ret(out, new A(*a)); else ret(out, materialize(a, a->dtype));
  OPS_CATCH
}
static std::vector<long> infer_shape(const A* a, const int64_t* shape, int32_t ndim) {
  std::vector<long> s(shape, shape + ndim);
  long known = 1;
  int neg = -1;
  for (int i = 0; i < ndim; ++i) {
    if (s[i] == -1) { if (neg >= 0) die("reshape: only one -1"); neg = i; }
    else known *= s[i];
  }
  const long n = a->numel();
  if (neg >= 0) { if (known == 0 || n % known != 0) die("reshape: size mismatch"); s[neg] = n / known; }
  else if (known != n) die("reshape: size mismatch");
  return s;
}
int32_t q3a_op_reshape(q3a_array** out, const q3a_array* a, const int64_t* shape, int32_t ndim, q3a_stream*) {
  OPS_TRY
  const auto s = infer_shape(a, shape, ndim);
  std::unique_ptr<A> c(is_contig(a) ? new A(*a) : materialize(a, a->dtype));
  View v = contig_view(s);
  v.offset = c->v.offset;
  ret(out, view_of(c.get(), v));
  OPS_CATCH
}
int32_t q3a_op_view(q3a_array** out, const q3a_array* a, const int64_t* shape, int32_t ndim) {
  OPS_TRY
  if (!is_contig(a)) die("view: the array is not contiguous (use reshape)");
  View v = contig_view(infer_shape(a, shape, ndim));
  v.offset = a->v.offset;
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_narrow(q3a_array** out, const q3a_array* a, int64_t dim, int64_t start, int64_t len) {
  OPS_TRY
  const int d = norm_dim(dim, a->v.nd);
  if (start < 0) start += a->v.shape[d];
  if (start < 0 || len < 0 || start + len > a->v.shape[d]) die("narrow: range out of bounds");
  View v = a->v;
  v.offset += start * v.stride[d];
  v.shape[d] = len;
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_select(q3a_array** out, const q3a_array* a, int64_t dim, int64_t index) {
  OPS_TRY
  const int d = norm_dim(dim, a->v.nd);
  if (index < 0) index += a->v.shape[d];
  if (index < 0 || index >= a->v.shape[d]) die("select: index out of range");
  View v{};
  v.nd = a->v.nd - 1;
  v.offset = a->v.offset + index * a->v.stride[d];
  for (int i = 0, j = 0; i < a->v.nd; ++i)
    if (i != d) { v.shape[j] = a->v.shape[i]; v.stride[j] = a->v.stride[i]; ++j; }
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_get(q3a_array** out, const q3a_array* a, int64_t index) { return q3a_op_select(out, a, 0, index); }
int32_t q3a_op_unsqueeze(q3a_array** out, const q3a_array* a, int64_t dim) {
  OPS_TRY
  const int d = norm_dim(dim, a->v.nd, true);
  if (a->v.nd + 1 > MAXD) die("too many dimensions");
  View v{};
  v.nd = a->v.nd + 1;
  v.offset = a->v.offset;
  for (int i = 0, j = 0; i < v.nd; ++i) {
    if (i == d) { v.shape[i] = 1; v.stride[i] = 1; }
    else { v.shape[i] = a->v.shape[j]; v.stride[i] = a->v.stride[j]; ++j; }
  }
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_squeeze_dim(q3a_array** out, const q3a_array* a, int64_t dim) {
  OPS_TRY
  const int d = norm_dim(dim, a->v.nd);
  if (a->v.shape[d] != 1) { ret(out, new A(*a)); return 0; }  // torch: no-op when the dimension is not 1
  View v{};
  v.nd = a->v.nd - 1;
  v.offset = a->v.offset;
  for (int i = 0, j = 0; i < a->v.nd; ++i)
    if (i != d) { v.shape[j] = a->v.shape[i]; v.stride[j] = a->v.stride[i]; ++j; }
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_transpose(q3a_array** out, const q3a_array* a, int64_t dim0, int64_t dim1) {
  OPS_TRY
  const int d0 = norm_dim(dim0, a->v.nd), d1 = norm_dim(dim1, a->v.nd);
  View v = a->v;
  std::swap(v.shape[d0], v.shape[d1]);
  std::swap(v.stride[d0], v.stride[d1]);
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_tr(q3a_array** out, const q3a_array* a) {
  if (a && a->v.nd != 2) { g_err = "tr: expects a 2-D array"; return 1; }
  return q3a_op_transpose(out, a, 0, 1);
}
int32_t q3a_op_permute(q3a_array** out, const q3a_array* a, const int64_t* dims, int32_t ndim) {
  OPS_TRY
  if (ndim != a->v.nd) die("permute: wrong number of dimensions");
  View v = a->v;
  std::vector<bool> seen(ndim, false);
  for (int i = 0; i < ndim; ++i) {
    const int d = norm_dim(dims[i], ndim);
    if (seen[d]) die("permute: repeated dimension");
    seen[d] = true;
    v.shape[i] = a->v.shape[d];
    v.stride[i] = a->v.stride[d];
  }
  ret(out, view_of(a, v));
  OPS_CATCH
}
int32_t q3a_op_expand(q3a_array** out, const q3a_array* a, const int64_t* size, int32_t ndim) {
  OPS_TRY
  if (ndim < a->v.nd || ndim > MAXD) die("expand: bad number of dimensions");
  View v{};
  v.nd = ndim;
  v.offset = a->v.offset;
  for (int i = 0; i < ndim; ++i) {
    const int da = a->v.nd - ndim + i;
    const long cur = da >= 0 ? a->v.shape[da] : 1;
    long want = size[i];
    if (want == -1) { if (da < 0) die("expand: -1 on a new leading dimension"); want = cur; }
    if (cur != want && cur != 1) die("expand: only size-1 dimensions expand");
    v.shape[i] = want;
    v.stride[i] = (da >= 0 && cur != 1) ? a->v.stride[da] : 0;
    if (cur == 1 && want == 1 && da >= 0) v.stride[i] = a->v.stride[da];
  }
  ret(out, view_of(a, v));
  OPS_CATCH
}

// ---- cat / stack / embedding ----
int32_t q3a_op_cat(q3a_array** out, const q3a_array* const* ts, int32_t n, int64_t dim, q3a_stream*) {
  OPS_TRY
  if (n < 1) die("cat: no tensors");
  const A* f = ts[0];
  const int d = norm_dim(dim, f->v.nd);
  std::vector<long> shape = shape_of(f);
  shape[d] = 0;
  for (int i = 0; i < n; ++i) {
    const A* t = ts[i];
    need_device(t, "cat");
    if (t->v.nd != f->v.nd || t->dtype != f->dtype || t->device != f->device) die("cat: mismatched rank / dtype / device");
    for (int k = 0; k < f->v.nd; ++k)
      if (k != d && t->v.shape[k] != f->v.shape[k]) die("cat: mismatched shapes");
    shape[d] += t->v.shape[d];
  }
  std::unique_ptr<A> r(make(shape, f->dtype, f->device));
  long at = 0;
  for (int i = 0; i < n; ++i) {
    const A* t = ts[i];
    View dv = r->v;
    dv.offset += at * dv.stride[d];
    dv.shape[d] = t->v.shape[d];
    k_copy_view(r->base(), r->dtype, dv, t->base(), t->dtype, t->v, t->numel(), sd(r.get()));
    at += t->v.shape[d];
  }
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_stack(q3a_array** out, const q3a_array* const* ts, int32_t n, int64_t dim, q3a_stream* s) {
  std::vector<q3a_array*> un((size_t)std::max(n, 0), nullptr);
  int32_t rc = 0;
  for (int i = 0; i < n && rc == 0; ++i) rc = q3a_op_unsqueeze(&un[i], ts[i], dim);
  if (rc == 0) rc = q3a_op_cat(out, un.data(), n, dim < 0 ? dim - 0 : dim, s);
  for (auto* u : un) q3a_array_free(u);
  return rc;
}
int32_t q3a_op_embedding(q3a_array** out, const q3a_array* w, const q3a_array* idx, q3a_stream*) {
  OPS_TRY
  need_device(w, "embedding");
  need_device(idx, "embedding");
  if (w->v.nd != 2) die("embedding: weight must be 2-D");
  if (idx->dtype != DT_I64) die("embedding: indices must be Int64");
  auto wc = f32c(w, "embedding");
  std::unique_ptr<A> ic(is_contig(idx) ? new A(*idx) : materialize(idx, DT_I64));
  std::vector<long> shape = shape_of(idx);
  shape.push_back(w->v.shape[1]);
  std::unique_ptr<A> r(make(shape, DT_F32, w->device));
  k_embedding(fpw(r.get()), fp(wc.get()), (const long long*)ic->data(), idx->numel(), (int)w->v.shape[1], sd(w));
  ret(out, r.release());
  OPS_CATCH
}

// ---- arithmetic ----
#define Q3A_BIN(NAME, OP) \
  int32_t q3a_op_##NAME(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream*) { OPS_TRY ret(out, binary(a, b, OP, #NAME)); OPS_CATCH }
Q3A_BIN(add, B_ADD)
Q3A_BIN(sub, B_SUB)
Q3A_BIN(mul, B_MUL)
Q3A_BIN(div, B_DIV)
Q3A_BIN(maximum, B_MAX)
#define Q3A_UN(NAME, OP) \
  int32_t q3a_op_##NAME(q3a_array** out, const q3a_array* a, q3a_stream*) { OPS_TRY ret(out, unary(a, OP, 0.f, #NAME)); OPS_CATCH }
#define Q3A_UNP(NAME, OP) \
  int32_t q3a_op_##NAME(q3a_array** out, const q3a_array* a, double p, q3a_stream*) { OPS_TRY ret(out, unary(a, OP, (float)p, #NAME)); OPS_CATCH }
Q3A_UN(neg, U_NEG)
Q3A_UN(square, U_SQUARE)
Q3A_UN(sqrt, U_SQRT)
Q3A_UN(rsqrt, U_RSQRT)
Q3A_UN(log10, U_LOG10)
Q3A_UN(sin, U_SIN)
Q3A_UN(cos, U_COS)
Q3A_UN(exp, U_EXP)
Q3A_UN(gelu, U_GELU)
Q3A_UN(silu, U_SILU)
Q3A_UNP(clamp_min, U_CLAMP_MIN)
Q3A_UNP(add_scalar, U_ADD_S)
Q3A_UNP(sub_scalar, U_SUB_S)
Q3A_UNP(mul_scalar, U_MUL_S)
Q3A_UNP(div_scalar, U_DIV_S)
Q3A_UNP(pow_scalar, U_POW_S)
int32_t q3a_op_abs(q3a_array** out, const q3a_array* a, q3a_stream*) {
  OPS_TRY
  if (a->dtype == DT_C64) {
    need_device(a, "abs");
    std::unique_ptr<A> c(is_contig(a) ? new A(*a) : materialize(a, DT_C64));
    std::unique_ptr<A> r(make(shape_of(a), DT_F32, a->device));
    k_complex_abs(fpw(r.get()), c->data(), r->numel(), sd(a));
    ret(out, r.release());
  } else {
    ret(out, unary(a, U_ABS, 0.f, "abs"));
  }
  OPS_CATCH
}
int32_t q3a_op_add_inplace(q3a_array* a, const q3a_array* b, q3a_stream*) {
  OPS_TRY
  std::unique_ptr<A> sum(binary(a, b, B_ADD, "add_"));
  if (shape_of(sum.get()) != shape_of(a)) die("add_: the result shape differs from the destination");
  if (a->dtype == DT_C64) die("add_: complex destination");
  k_copy_view(a->base(), a->dtype, a->v, sum->base(), DT_F32, sum->v, a->numel(), sd(a));
  OPS_CATCH
}
int32_t q3a_op_fill_inplace(q3a_array* a, double v, q3a_stream*) {
  OPS_TRY
  need_device(a, "fill_");
  k_fill(a->base(), a->dtype, a->v, a->numel(), v, sd(a));
  OPS_CATCH
}

int32_t q3a_op_matmul(q3a_array** out, const q3a_array* a0, const q3a_array* b0, q3a_stream*) {
  OPS_TRY
  need_device(a0, "matmul");
  need_device(b0, "matmul");
  // 1-D operands: promote, remember to squeeze
  std::unique_ptr<A> ap(new A(*a0)), bp(new A(*b0));
  bool sq_m = false, sq_n = false;
  if (ap->v.nd == 1) { q3a_array* t; if (q3a_op_unsqueeze(&t, ap.get(), 0)) die(g_err); ap.reset(t); sq_m = true; }
  if (bp->v.nd == 1) { q3a_array* t; if (q3a_op_unsqueeze(&t, bp.get(), 1)) die(g_err); bp.reset(t); sq_n = true; }
  const int na = ap->v.nd, nb = bp->v.nd;
  const long M = ap->v.shape[na - 2], K = ap->v.shape[na - 1], Kb = bp->v.shape[nb - 2], N = bp->v.shape[nb - 1];
  if (K != Kb) die("matmul: inner dimensions differ");
  // batch shape by broadcasting the leading dimensions
  const int nbd = std::max(na, nb) - 2;
  std::vector<long> bshape(nbd);
  for (int i = 0; i < nbd; ++i) {
    const int da = na - 2 - nbd + i, db = nb - 2 - nbd + i;
    const long xa = da >= 0 ? ap->v.shape[da] : 1, xb = db >= 0 ? bp->v.shape[db] : 1;
    if (xa != xb && xa != 1 && xb != 1) die("matmul: batch dimensions cannot be broadcast");
    bshape[i] = xa == 1 ? xb : xa;
  }
  long batch = 1;
  for (long x : bshape) batch *= x;
  auto expand_to = [&](std::unique_ptr<A>& t, long r, long c, bool& shared) {
    // returns a contiguous F32 array of shape bshape + [r, c], or [r, c] with `shared` when t has no batch extent
    long tb = 1;
    for (int i = 0; i < t->v.nd - 2; ++i) tb *= t->v.shape[i];
    shared = tb == 1 && batch > 1;
    if (tb == 1) {
      q3a_array* m;
      const int64_t rs[2] = {r, c};
      if (q3a_op_reshape(&m, t.get(), rs, 2, nullptr)) die(g_err);
      std::unique_ptr<A> mm(m);
      return f32c(mm.get(), "matmul");
    }
    std::vector<int64_t> full(bshape.begin(), bshape.end());
    full.push_back(r);
    full.push_back(c);
    q3a_array* e;
    if (q3a_op_expand(&e, t.get(), full.data(), (int32_t)full.size())) die(g_err);
    std::unique_ptr<A> ee(e);
    return std::unique_ptr<A>(ee->dtype == DT_F32 && is_contig(ee.get()) ? new A(*ee) : materialize(ee.get(), DT_F32));
  };
  bool sha = false, shb = false, tb = false;
  auto ac = expand_to(ap, M, K, sha);
  std::unique_ptr<A> bc;
  if (nb == 2 && bp->dtype == DT_F32 && bp->v.stride[0] == 1 && bp->v.stride[1] == K && N > 1) {
    // `weight.tr()` of a contiguous [N][K] matrix (Linear::forward): read the weight in place
    bc.reset(new A(*bp));
    tb = true;
    shb = batch > 1;
  } else {
    bc = expand_to(bp, K, N, shb);
  }
  std::vector<long> oshape(bshape);
  oshape.push_back(M);
  oshape.push_back(N);
  std::unique_ptr<A> r(make(oshape, DT_F32, a0->device));
  k_matmul(fpw(r.get()), fp(ac.get()), fp(bc.get()), (int)batch, (int)M, (int)N, (int)K, sha ? 0 : M * K, shb ? 0 : K * N, tb, sd(a0));
  A* res = r.release();
  if (sq_m || sq_n) {
    std::vector<int64_t> fs;
    for (int i = 0; i < res->v.nd; ++i) {
      if (sq_m && i == res->v.nd - 2) continue;
      if (sq_n && i == res->v.nd - 1) continue;
      fs.push_back(res->v.shape[i]);
    }
    q3a_array* t;
    const int32_t rc = q3a_op_reshape(&t, res, fs.data(), (int32_t)fs.size(), nullptr);
    delete res;
    if (rc) die(g_err);
    res = t;
  }
  ret(out, res);
  OPS_CATCH
}

// ---- reductions over one dimension: move it last, make contiguous, one workgroup per row ----
static std::unique_ptr<A> rows_last(const A* a, int dim, const char* op) {
  need_device(a, op);
  std::unique_ptr<A> v(new A(*a));
  v->v = move_dim_last(a->v, dim);
  return std::unique_ptr<A>(v->dtype == DT_F32 && is_contig(v.get()) ? new A(*v) : materialize(v.get(), DT_F32));
}
int32_t q3a_op_softmax(q3a_array** out, const q3a_array* a, int64_t dim, q3a_stream*) {
  OPS_TRY
  const int d = norm_dim(dim, a->v.nd);
  auto x = rows_last(a, d, "softmax");
  const long D = a->v.shape[d], rows = D ? a->numel() / D : 0;
  std::unique_ptr<A> y(make(shape_of(x.get()), DT_F32, a->device));
  k_softmax_rows(fpw(y.get()), fp(x.get()), rows, (int)D, sd(a));
  if (d == a->v.nd - 1) { ret(out, y.release()); return 0; }
  // move the last dimension back to position d (a view; consumers make it contiguous when they need to)
  View v = y->v, r = y->v;
  for (int i = 0, j = 0; i < a->v.nd; ++i) {
    if (i == d) { r.shape[i] = v.shape[a->v.nd - 1]; r.stride[i] = v.stride[a->v.nd - 1]; }
    else { r.shape[i] = v.shape[j]; r.stride[i] = v.stride[j]; ++j; }
  }
  ret(out, view_of(y.get(), r));
  OPS_CATCH
}
int32_t q3a_op_mean_dim(q3a_array** out, const q3a_array* a, const int64_t* dims, int32_t ndims, int32_t keepdim, q3a_stream*) {
  OPS_TRY
  need_device(a, "mean_dim");
  std::vector<bool> red(a->v.nd, false);
  for (int i = 0; i < ndims; ++i) red[norm_dim(dims[i], a->v.nd)] = true;
  View p = a->v;
  int j = 0;
  long D = 1;
  std::vector<long> oshape;
  for (int d = 0; d < a->v.nd; ++d)
    if (!red[d]) { p.shape[j] = a->v.shape[d]; p.stride[j] = a->v.stride[d]; ++j; oshape.push_back(a->v.shape[d]); }
    else if (keepdim) oshape.push_back(1);
  for (int d = 0; d < a->v.nd; ++d)
    if (red[d]) { p.shape[j] = a->v.shape[d]; p.stride[j] = a->v.stride[d]; ++j; D *= a->v.shape[d]; }
  if (keepdim) {  // keepdim output shape keeps the order of the input dimensions
    oshape.clear();
    for (int d = 0; d < a->v.nd; ++d) oshape.push_back(red[d] ? 1 : a->v.shape[d]);
  }
  std::unique_ptr<A> pv(new A(*a));
  pv->v = p;
  std::unique_ptr<A> x(pv->dtype == DT_F32 && is_contig(pv.get()) ? new A(*pv) : materialize(pv.get(), DT_F32));
  std::unique_ptr<A> r(make(oshape, DT_F32, a->device));
  k_mean_rows(fpw(r.get()), fp(x.get()), D ? a->numel() / D : 0, (int)D, sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_max(q3a_array** out, const q3a_array* a, q3a_stream*) {
  OPS_TRY
  auto x = f32c(a, "max");
  if (a->numel() == 0) die("max: empty array");
  std::unique_ptr<A> r(make({}, DT_F32, a->device));
  k_argmax_rows(nullptr, fpw(r.get()), fp(x.get()), 1, (int)a->numel(), sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_argmax(q3a_array** out, const q3a_array* a, int64_t dim, int32_t keepdim, q3a_stream*) {
  OPS_TRY
  const int d = norm_dim(dim, a->v.nd);
  auto x = rows_last(a, d, "argmax");
  const long D = a->v.shape[d];
  if (D == 0) die("argmax: empty dimension");
  std::vector<long> oshape;
  for (int i = 0; i < a->v.nd; ++i) {
    if (i == d) { if (keepdim) oshape.push_back(1); }
    else oshape.push_back(a->v.shape[i]);
  }
  std::unique_ptr<A> r(make(oshape, DT_I64, a->device));
  k_argmax_rows((long long*)r->data(), nullptr, fp(x.get()), a->numel() / D, (int)D, sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_triu(q3a_array** out, const q3a_array* a, int64_t diagonal, q3a_stream*) {
  OPS_TRY
  if (a->v.nd < 2) die("triu: needs at least 2 dimensions");
  auto x = f32c(a, "triu");
  std::unique_ptr<A> r(make(shape_of(a), DT_F32, a->device));
  k_triu(fpw(r.get()), fp(x.get()), r->numel(), (int)a->v.shape[a->v.nd - 2], (int)a->v.shape[a->v.nd - 1], (long)diagonal, sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_slice_scatter(q3a_array** out, const q3a_array* a, const q3a_array* src, int64_t dim, int64_t start, int64_t end,
                             int64_t step, q3a_stream*) {
  OPS_TRY
  need_device(a, "slice_scatter");
  need_device(src, "slice_scatter");
  const int d = norm_dim(dim, a->v.nd);
  const long L = a->v.shape[d];
  if (step < 1) die("slice_scatter: step must be positive");
  if (start < 0) start += L;
  if (end < 0) end += L;
  start = std::min<long>(std::max<long>(start, 0), L);
  end = std::min<long>(std::max<long>(end, start), L);
  const long len = (end - start + step - 1) / step;
  std::unique_ptr<A> r(materialize(a, a->dtype));
  View dv = r->v;
  dv.offset += start * dv.stride[d];
  dv.shape[d] = len;
  dv.stride[d] *= step;
  if (src->v.nd != a->v.nd) die("slice_scatter: src rank differs");
  for (int i = 0; i < a->v.nd; ++i)
    if (src->v.shape[i] != dv.shape[i]) die("slice_scatter: src shape does not match the slice");
  k_copy_view(r->base(), r->dtype, dv, src->base(), src->dtype, src->v, src->numel(), sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_layer_norm(q3a_array** out, const q3a_array* a, const int64_t* ns, int32_t n, const q3a_array* weight, const q3a_array* bias,
                          double eps, q3a_stream*) {
  OPS_TRY
  if (n != 1 || a->v.nd < 1 || ns[0] != a->v.shape[a->v.nd - 1]) die("layer_norm: normalized_shape must be the last dimension");
  auto x = f32c(a, "layer_norm");
  const int D = (int)ns[0];
  std::unique_ptr<A> wc, bc;
  if (weight) wc = f32c(weight, "layer_norm");
  if (bias) bc = f32c(bias, "layer_norm");
  std::unique_ptr<A> r(make(shape_of(a), DT_F32, a->device));
  const long rows = D ? a->numel() / D : 0;
  if (wc && bc && D % 4 == 0 && D <= 2048) {  // the engine's LayerNorm kernel (k_norm.hip)
    const char* e = q3a::launch_layernorm(fp(x.get()), fp(wc.get()), fp(bc.get()), fpw(r.get()), (int)rows, D, (float)eps, sd(a), nullptr);
    if (e) die(e);
  } else {
    k_layernorm_rows(fpw(r.get()), fp(x.get()), wc ? fp(wc.get()) : nullptr, bc ? fp(bc.get()) : nullptr, rows, D, (float)eps, sd(a));
  }
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_conv2d(q3a_array** out, const q3a_array* a, const q3a_array* weight, const q3a_array* bias, const int64_t* stride,
                      const int64_t* padding, const int64_t* dilation, int64_t groups, q3a_stream*) {
  OPS_TRY
  if (groups != 1) die("conv2d: groups must be 1");
  if (a->v.nd != 4 || weight->v.nd != 4) die("conv2d: NCHW input and OIHW weight expected");
  auto x = f32c(a, "conv2d");
  auto w = f32c(weight, "conv2d");
  std::unique_ptr<A> b;
  if (bias) b = f32c(bias, "conv2d");
  ConvDims d{};
  d.N = (int)a->v.shape[0]; d.Ci = (int)a->v.shape[1]; d.H = (int)a->v.shape[2]; d.W = (int)a->v.shape[3];
  d.Co = (int)weight->v.shape[0]; d.KH = (int)weight->v.shape[2]; d.KW = (int)weight->v.shape[3];
  if (weight->v.shape[1] != d.Ci) die("conv2d: channel mismatch");
  d.sh = (int)stride[0]; d.sw = (int)stride[1]; d.ph = (int)padding[0]; d.pw = (int)padding[1]; d.dh = (int)dilation[0]; d.dw = (int)dilation[1];
  d.OH = (d.H + 2 * d.ph - d.dh * (d.KH - 1) - 1) / d.sh + 1;
  d.OW = (d.W + 2 * d.pw - d.dw * (d.KW - 1) - 1) / d.sw + 1;
  std::unique_ptr<A> r(make({d.N, d.Co, d.OH, d.OW}, DT_F32, a->device));
  k_conv2d(fpw(r.get()), fp(x.get()), fp(w.get()), b ? fp(b.get()) : nullptr, d, sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_reflection_pad1d(q3a_array** out, const q3a_array* a, const int64_t* pad, q3a_stream*) {
  OPS_TRY
  if (a->v.nd < 1) die("reflection_pad1d: needs a dimension");
  auto x = f32c(a, "reflection_pad1d");
  const long n = a->v.shape[a->v.nd - 1];
  if (pad[0] < 0 || pad[1] < 0 || pad[0] >= n || pad[1] >= n) die("reflection_pad1d: padding must be smaller than the input");
  std::vector<long> shape = shape_of(a);
  shape.back() = n + pad[0] + pad[1];
  std::unique_ptr<A> r(make(shape, DT_F32, a->device));
  k_reflect_pad(fpw(r.get()), fp(x.get()), n ? a->numel() / n : 0, n, pad[0], pad[1], sd(a));
  ret(out, r.release());
  OPS_CATCH
}
int32_t q3a_op_stft(q3a_array** out, const q3a_array* a, int64_t n_fft, int64_t hop, int64_t win_length, const q3a_array* window,
                    int32_t normalized, int32_t onesided, int32_t return_complex, q3a_stream*) {
  OPS_TRY
  if (a->v.nd != 1) die("stft: 1-D input expected (the reference removes the batch dimensions first, src/mel.rs:63-65)");
  if (win_length != n_fft || !window || window->numel() != n_fft) die("stft: win_length must equal n_fft and a window is required");
  auto x = f32c(a, "stft");
  auto w = f32c(window, "stft");
  const long L = a->v.shape[0];
  if (L < n_fft) die("stft: input shorter than n_fft");
  const int n_frames = (int)(1 + (L - n_fft) / hop), n_freq = onesided ? (int)(n_fft / 2 + 1) : (int)n_fft;
  DftTables t;
  {
    auto key = std::make_tuple((int)n_fft, n_freq, a->device);
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = g_dft.find(key);
    if (it == g_dft.end()) {
      lk.unlock();
      std::vector<float> c((size_t)n_freq * n_fft), s((size_t)n_freq * n_fft);
      for (int k = 0; k < n_freq; ++k)
        for (int tt = 0; tt < n_fft; ++tt) {
          const double ang = 2.0 * 3.14159265358979323846 * (double)((long)k * tt % n_fft) / (double)n_fft;
          c[(size_t)k * n_fft + tt] = (float)std::cos(ang);
          s[(size_t)k * n_fft + tt] = (float)std::sin(ang);
        }
      t.ct = alloc(c.size() * 4, a->device);
      t.st = alloc(s.size() * 4, a->device);
      h2d(t.ct->p, c.data(), c.size() * 4, a->device);
      h2d(t.st->p, s.data(), s.size() * 4, a->device);
      lk.lock();
      g_dft[key] = t;
    } else {
      t = it->second;
    }
  }
  std::unique_ptr<A> r(return_complex ? make({n_freq, n_frames}, DT_C64, a->device) : make({n_freq, n_frames, 2}, DT_F32, a->device));
  k_stft(r->base(), fp(x.get()), fp(w.get()), (const float*)t.ct->p, (const float*)t.st->p, (int)n_fft, (int)hop, n_frames, n_freq,
         normalized ? 1.0f / std::sqrt((float)n_fft) : 1.0f, sd(a));
  ret(out, r.release());
  OPS_CATCH
}

// ---- dtype / device / extraction ----
int32_t q3a_op_to_dtype(q3a_array** out, const q3a_array* a, int32_t dtype, q3a_stream*) {
  OPS_TRY
  if (dtype < 0 || dtype > DT_BOOL) die("to_dtype: bad dtype");
  if (dtype == a->dtype) ret(out, new A(*a)); else ret(out, materialize(a, dtype));
  OPS_CATCH
}
int32_t q3a_op_to_device(q3a_array** out, const q3a_array* a, int32_t device, q3a_stream*) {
  OPS_TRY
  if (device == a->device) { ret(out, new A(*a)); return 0; }
  std::unique_ptr<A> c(is_contig(a) ? new A(*a) : materialize(a, a->dtype));
  std::unique_ptr<A> r(make(shape_of(a), a->dtype, device));
  const size_t bytes = (size_t)a->numel() * dtype_size(a->dtype);
  if (bytes) {
    if (a->device == Q3A_CPU) {
      h2d(r->base(), c->data(), bytes, device);
    } else if (device == Q3A_CPU) {
      OHIP(hipStreamSynchronize(sd(a)));
      OHIP(hipMemcpy(r->base(), c->data(), bytes, hipMemcpyDeviceToHost));
    } else {
      OHIP(hipStreamSynchronize(sd(a)));
      // the destination block may be a recycled one whose last user is still queued on the DESTINATION device's
      // (non-blocking) ops stream: drain that stream before the peer copy lands in it
      OHIP(hipSetDevice(device));
      OHIP(hipStreamSynchronize(stream_of(device)));
      OHIP(hipMemcpyPeer(r->base(), device, c->data(), a->device, bytes));
    }
  }
  ret(out, r.release());
  OPS_CATCH
}
static double element_as_double(const q3a_array* a, const int64_t* indices, int32_t n) {
  if (n != a->v.nd) die("value(): one index per dimension expected");
  long off = a->v.offset;
  for (int d = 0; d < n; ++d) {
    long i = indices[d];
    if (i < 0) i += a->v.shape[d];
    if (i < 0 || i >= a->v.shape[d]) die("value(): index out of range");
    off += i * a->v.stride[d];
  }
  uint8_t buf[8] = {0};
  const int es = dtype_size(a->dtype);
  if (a->device == Q3A_CPU) memcpy(buf, (const uint8_t*)a->base() + (size_t)off * es, es);
  else {
    OHIP(hipStreamSynchronize(sd(a)));
    OHIP(hipMemcpy(buf, (const uint8_t*)a->base() + (size_t)off * es, es, hipMemcpyDeviceToHost));
  }
  switch (a->dtype) {
    case DT_F32: { float f; memcpy(&f, buf, 4); return f; }
    case DT_I64: { int64_t v; memcpy(&v, buf, 8); return (double)v; }
    case DT_I32: { int32_t v; memcpy(&v, buf, 4); return v; }
    case DT_BOOL: return buf[0] ? 1.0 : 0.0;
    case DT_BF16: { uint32_t u = (uint32_t)(buf[0] | (buf[1] << 8)) << 16; float f; memcpy(&f, &u, 4); return f; }
    default: die("value(): unsupported dtype");
  }
}
int32_t q3a_array_int64_value(const q3a_array* a, const int64_t* indices, int32_t n, int64_t* value) {
  OPS_TRY
  if (a->dtype == DT_I64) {  // exact for the whole int64 range
    q3a_array* e = nullptr;
    const double d = element_as_double(a, indices, n);
    (void)e;
    *value = (int64_t)d;
    if (std::fabs(d) > 9007199254740992.0) die("int64_value: magnitude beyond 2^53 is not supported");
  } else {
    *value = (int64_t)element_as_double(a, indices, n);
  }
  OPS_CATCH
}
int32_t q3a_array_f64_value(const q3a_array* a, const int64_t* indices, int32_t n, double* value) {
  OPS_TRY
  *value = element_as_double(a, indices, n);
  OPS_CATCH
}
int32_t q3a_array_to_vec_f32(const q3a_array* a, float* dst, int64_t cap) {
  OPS_TRY
  if (cap < a->numel()) die("to_vec_f32: destination too small");
  std::unique_ptr<A> c(a->dtype == DT_F32 && is_contig(a) ? new A(*a) : materialize(a, DT_F32));
  const size_t bytes = (size_t)a->numel() * 4;
  if (!bytes) return 0;
  if (c->device == Q3A_CPU) memcpy(dst, c->data(), bytes);
  else {
    OHIP(hipStreamSynchronize(sd(c.get())));
    OHIP(hipMemcpy(dst, c->data(), bytes, hipMemcpyDeviceToHost));
  }
  OPS_CATCH
}

}  // extern "C"
