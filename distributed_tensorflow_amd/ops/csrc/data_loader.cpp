// Host-side input pipeline in C++ (module `_dtf_data`): the parts of the reference's tf.data
// graph that are per-byte work and would otherwise run in the Python interpreter.
//
//  * CifarLoader   FixedLengthRecordDataset(3073 B) -> shard -> shuffle(whole set) -> repeat ->
//                  map(pad 4 / random crop 32 / random flip / per-image standardisation) ->
//                  batch, as `cifar_preprocessing.input_fn` does (`cifar_preprocessing.py:43-164`
//                  of the reference), with a pool of worker threads filling a caller-provided
//                  (pinned) batch buffer.
//  * TFRecord      framing reader with CRC-32C verification (slicing-by-8) and an index of
//                  record offsets, for the ImageNet shards (`imagenet_preprocessing.py:259-323`).
//
// No GPU code here; the buffers these functions fill are pinned host tensors that the device
// prefetcher copies on its own stream.
#include "jpeg_region.h"
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace py = pybind11;

namespace {

// ---------------------------------------------------------------------------------------------
// CRC-32C (Castagnoli), slicing-by-8
// ---------------------------------------------------------------------------------------------
struct Crc32cTable {
  uint32_t t[8][256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
const Crc32cTable& table() {
  static const Crc32cTable tb;
  return tb;
}

uint32_t crc32c(const uint8_t* p, size_t n, uint32_t crc = 0) {
  const auto& T = table().t;
  crc = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
          T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

uint32_t masked_crc32c(const uint8_t* p, size_t n) {
  const uint32_t c = crc32c(p, n);
  return (((c >> 15) | (c << 17)) + 0xA282EAD8u);
}

// ---------------------------------------------------------------------------------------------
// TFRecord framing
// ---------------------------------------------------------------------------------------------
std::vector<uint8_t> read_file(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> buf((size_t)std::max(n, 0L));
  if (n > 0 && std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) {
    std::fclose(f);
    throw std::runtime_error("short read on " + path);
  }
  std::fclose(f);
  return buf;
}

// Returns every record payload of a TFRecord file; throws on a framing or checksum error.
std::vector<py::bytes> tfrecord_read_all(const std::string& path, bool verify) {
  std::vector<uint8_t> buf;
  {
    py::gil_scoped_release nogil;
    buf = read_file(path);
  }
  std::vector<std::pair<size_t, size_t>> spans;
  {
    py::gil_scoped_release nogil;
    size_t pos = 0;
    while (pos < buf.size()) {
      if (pos + 12 > buf.size()) throw std::runtime_error("truncated record header in " + path);
      uint64_t len;
      uint32_t hcrc;
      std::memcpy(&len, &buf[pos], 8);
      std::memcpy(&hcrc, &buf[pos + 8], 4);
      if (verify && hcrc != masked_crc32c(&buf[pos], 8))
        throw std::runtime_error("corrupt record length checksum in " + path);
      if (len > buf.size() || pos + 12 + len + 4 > buf.size())
        throw std::runtime_error("truncated record in " + path);
      uint32_t dcrc;
      std::memcpy(&dcrc, &buf[pos + 12 + len], 4);
      if (verify && dcrc != masked_crc32c(&buf[pos + 12], (size_t)len))
        throw std::runtime_error("corrupt record data checksum in " + path);
      spans.emplace_back(pos + 12, (size_t)len);
      pos += 12 + len + 4;
    }
  }
  std::vector<py::bytes> out;
  out.reserve(spans.size());
  for (auto& s : spans) out.emplace_back(reinterpret_cast<const char*>(&buf[s.first]), s.second);
  return out;
}

// ---------------------------------------------------------------------------------------------
// A small fixed thread pool running `fn(i)` for i in [0, n)
// ---------------------------------------------------------------------------------------------
class Pool {
 public:
  explicit Pool(int threads) {
    for (int t = 0; t < threads; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (workers_.empty() || n <= 1) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      n_ = n;
      next_ = 0;
      pending_ = n;
      ++generation_;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return stop_ || (generation_ != seen && next_ < n_); });
      if (stop_) return;
      while (next_ < n_) {
        const int i = next_++;
        const std::function<void(int)>* fn = fn_;
        lk.unlock();
        (*fn)(i);
        lk.lock();
        if (--pending_ == 0) done_.notify_all();
      }
      seen = generation_;
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, next_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

// ---------------------------------------------------------------------------------------------
// CIFAR-10
// ---------------------------------------------------------------------------------------------
constexpr int kH = 32, kW = 32, kC = 3, kRecord = 1 + kH * kW * kC, kClasses = 10;

class CifarLoader {
 public:
  CifarLoader(const std::vector<std::string>& files, bool is_training, int batch_size,
              int shard_index, int num_shards, uint64_t seed, int num_threads, bool drop_remainder,
              int num_epochs)
      : training_(is_training), batch_(batch_size), drop_(drop_remainder), epochs_(num_epochs),
        rng_(seed * 7919u + (uint64_t)shard_index * 131u + 17u),
        pool_(std::max(num_threads, 0)) {
    if (batch_size <= 0) throw std::invalid_argument("batch_size must be positive");
    if (num_shards < 1 || shard_index < 0 || shard_index >= num_shards)
      throw std::invalid_argument("bad shard specification");
    size_t global = 0;
    for (const auto& f : files) {
      std::vector<uint8_t> buf = read_file(f);
      const size_t n = buf.size() / kRecord;
      for (size_t r = 0; r < n; ++r, ++global) {
        if ((int)(global % (size_t)num_shards) != shard_index) continue;   // shard by record
        data_.insert(data_.end(), buf.begin() + r * kRecord, buf.begin() + (r + 1) * kRecord);
      }
    }
    count_ = data_.size() / kRecord;
    if (count_ == 0) throw std::runtime_error("no CIFAR records for this shard");
    order_.resize(count_);
    for (size_t i = 0; i < count_; ++i) order_[i] = (uint32_t)i;
    start_epoch();
  }

  size_t num_records() const { return count_; }

  // Fills images [B,32,32,3] float32 and labels [B,10] float32 (one-hot); returns the number of
  // samples written (0 at the end of the last epoch).
  int next_batch(py::buffer images, py::buffer labels) {
    py::buffer_info bi = images.request(true), bl = labels.request(true);
    if (bi.itemsize != 4 || bl.itemsize != 4 || bi.size < (py::ssize_t)batch_ * kH * kW * kC ||
        bl.size < (py::ssize_t)batch_ * kClasses)
      throw std::invalid_argument("next_batch needs float32 buffers of [B,32,32,3] and [B,10]");
    float* img = static_cast<float*>(bi.ptr);
    float* lab = static_cast<float*>(bl.ptr);
    std::vector<uint32_t> idx;
    std::vector<uint32_t> aug;      // per sample: crop y | crop x << 8 | flip << 16
    // shuffle -> repeat -> batch: a batch may span two epochs; only the tail of the last epoch
    // can be short (and is dropped with drop_remainder)
    while ((int)idx.size() < batch_) {
      if (cursor_ == count_) {
        if (epochs_ > 0 && epoch_ >= epochs_) {
          if (drop_) {
            idx.clear();
            aug.clear();
          }
          break;
        }
        start_epoch();
        continue;
      }
      idx.push_back(order_[cursor_++]);
      uint32_t a = 0;
      if (training_)
        a = (uint32_t)(rng_() % 9) | ((uint32_t)(rng_() % 9) << 8) | ((uint32_t)(rng_() & 1) << 16);
      aug.push_back(a);
    }
    const int n = (int)idx.size();
    if (n == 0) return 0;
    {
      py::gil_scoped_release nogil;
      const int chunks = std::min(n, 64);
      pool_.parallel_for(chunks, [&](int c) {
        for (int s = c; s < n; s += chunks)
          sample(&data_[(size_t)idx[s] * kRecord], aug[s], img + (size_t)s * kH * kW * kC,
                 lab + (size_t)s * kClasses);
      });
    }
    return n;
  }

 private:
  void start_epoch() {
    ++epoch_;
    cursor_ = 0;
    if (training_) std::shuffle(order_.begin(), order_.end(), rng_);
  }

  void sample(const uint8_t* rec, uint32_t aug, float* out, float* lab) const {
    for (int k = 0; k < kClasses; ++k) lab[k] = 0.f;
    lab[std::min<int>(rec[0], kClasses - 1)] = 1.f;
    const uint8_t* px = rec + 1;                 // CHW planes
    const int oy = training_ ? (int)(aug & 0xFF) - 4 : 0;
    const int ox = training_ ? (int)((aug >> 8) & 0xFF) - 4 : 0;
    const bool flip = training_ && ((aug >> 16) & 1);
    double sum = 0.0, sq = 0.0;
    for (int y = 0; y < kH; ++y) {
      const int sy = y + oy;
      for (int x = 0; x < kW; ++x) {
        const int xx = flip ? kW - 1 - x : x;
        const int sx = xx + ox;
        const bool in = sy >= 0 && sy < kH && sx >= 0 && sx < kW;
        for (int c = 0; c < kC; ++c) {
          const float v = in ? (float)px[(c * kH + sy) * kW + sx] : 0.f;
          out[(y * kW + x) * kC + c] = v;
          sum += v;
          sq += (double)v * v;
        }
      }
    }
    // tf.image.per_image_standardization: (x - mean) / max(stddev, 1/sqrt(N))
    const double nel = (double)kH * kW * kC;
    const double mean = sum / nel;
    const double var = std::max(sq / nel - mean * mean, 0.0);
    const float inv = (float)(1.0 / std::max(std::sqrt(var), 1.0 / std::sqrt(nel)));
    const float m = (float)mean;
    for (int i = 0; i < kH * kW * kC; ++i) out[i] = (out[i] - m) * inv;
  }

  bool training_;
  int batch_;
  bool drop_;
  int epochs_;                    // <= 0: repeat forever
  int epoch_ = 0;
  size_t cursor_ = 0, count_ = 0;
  std::vector<uint8_t> data_;
  std::vector<uint32_t> order_;
  std::mt19937_64 rng_;
  Pool pool_;
};

}  // namespace

// Bilinear resize of a crop of a decoded uint8 HWC image to float32, fused with the horizontal
// flip and the per-channel mean subtraction of the ImageNet pipeline.  Sampling follows the
// legacy TensorFlow rule the reference relies on (`tf.compat.v1.image.resize(...,
// align_corners=False)`, `imagenet_preprocessing.py:483-500`): source coordinate = output index
// * (input size / output size), the two nearest samples per axis, no antialiasing.
//   out[i][j] = resized(src[y:y+h, x:x+w] -> [rh, rw])[i + top][j + left] - mean
// with the flip applied to the output columns.
// `origin_y`, `origin_x`: `src` holds only the part of the image that starts there (a region
// decode); the box and the sampling rule keep referring to the whole image, and every sample the
// output window needs must lie inside the part.
py::array_t<float> crop_resize_bilinear(
    py::array_t<uint8_t, py::array::c_style | py::array::forcecast> src, int y, int x, int h, int w,
    int rh, int rw, int top, int left, int out_h, int out_w, bool flip,
    const std::vector<float>& means, int origin_y, int origin_x) {
  if (src.ndim() != 3) throw std::invalid_argument("crop_resize_bilinear: image must be [H, W, C]");
  const int H = (int)src.shape(0), W = (int)src.shape(1), C = (int)src.shape(2);
  if (y < 0 || x < 0 || h <= 0 || w <= 0 || origin_y < 0 || origin_x < 0)
    throw std::invalid_argument("crop_resize_bilinear: crop box outside the image");
  if (rh <= 0 || rw <= 0 || top < 0 || left < 0 || top + out_h > rh || left + out_w > rw ||
      out_h <= 0 || out_w <= 0)
    throw std::invalid_argument("crop_resize_bilinear: output window outside the resized image");
  {
    // first / last source row and column the window samples (the rule below)
    const float sy = (float)h / (float)rh, sx = (float)w / (float)rw;
    const int r_first = y + std::min((int)((float)top * sy), h - 1);
    const int r_last = y + std::min((int)((float)(top + out_h - 1) * sy) + 1, h - 1);
    const int c_first = x + std::min((int)((float)left * sx), w - 1);
    const int c_last = x + std::min((int)((float)(left + out_w - 1) * sx) + 1, w - 1);
    if (r_first < origin_y || r_last >= origin_y + H || c_first < origin_x || c_last >= origin_x + W)
      throw std::invalid_argument("crop_resize_bilinear: crop box outside the image");
  }
  if (!means.empty() && (int)means.size() != C)
    throw std::invalid_argument("crop_resize_bilinear: one mean per channel");
  py::array_t<float> out({out_h, out_w, C});
  const uint8_t* in = src.data();
  float* o = out.mutable_data();
  {
    py::gil_scoped_release nogil;
    const float sy = (float)h / (float)rh, sx = (float)w / (float)rw;
    // columns: element offsets (relative to the crop's first column) and weights, once per image
    std::vector<int> x0(out_w), x1(out_w);
    std::vector<float> fx(out_w);
    for (int j = 0; j < out_w; ++j) {
      const float f = (float)(j + left) * sx;
      const int lo = (int)f;
      x0[j] = std::min(lo, w - 1) * C;
      x1[j] = std::min(lo + 1, w - 1) * C;
      fx[j] = f - (float)lo;
    }
    std::vector<float> mean(C, 0.f);
    for (int k = 0; k < C && k < (int)means.size(); ++k) mean[k] = means[k];
    // per output row: blend the two source rows over the crop's width (a contiguous, vectorisable
    // sweep), then pick / blend the two neighbouring columns of every output pixel
    std::vector<float> line((size_t)w * C);
    // columns of the crop the window touches: only those are blended (and need to exist in `src`)
    int e_lo = x0[0], e_hi = x1[0];
    for (int j = 0; j < out_w; ++j) { e_lo = std::min(e_lo, x0[j]); e_hi = std::max(e_hi, x1[j]); }
    e_hi += C;
    for (int i = 0; i < out_h; ++i) {
      const float f = (float)(i + top) * sy;
      const int lo = (int)f;
      const float fy = f - (float)lo;
      const uint8_t* r0 = in + ((ptrdiff_t)(y + std::min(lo, h - 1) - origin_y) * W + (x - origin_x)) * C;
      const uint8_t* r1 = in + ((ptrdiff_t)(y + std::min(lo + 1, h - 1) - origin_y) * W + (x - origin_x)) * C;
      float* t = line.data();
      for (int e = e_lo; e < e_hi; ++e) {
        const float a = (float)r0[e];
        t[e] = a + ((float)r1[e] - a) * fy;
      }
      float* orow = o + (size_t)i * out_w * C;
      for (int j = 0; j < out_w; ++j) {
        const float* a = t + x0[j];
        const float* b = t + x1[j];
        float* dst = orow + (size_t)(flip ? out_w - 1 - j : j) * C;
        const float wx = fx[j];
        for (int k = 0; k < C; ++k) dst[k] = a[k] + (b[k] - a[k]) * wx - mean[k];
      }
    }
  }
  return out;
}

PYBIND11_MODULE(_dtf_data, m) {
  m.doc() = "native host-side input pipeline (CIFAR-10 loader, TFRecord reader, CRC-32C)";
  m.def("crc32c", [](py::bytes b, uint32_t crc) {
    const std::string s = b;
    return crc32c(reinterpret_cast<const uint8_t*>(s.data()), s.size(), crc);
  }, py::arg("data"), py::arg("crc") = 0);
  m.def("masked_crc32c", [](py::bytes b) {
    const std::string s = b;
    return masked_crc32c(reinterpret_cast<const uint8_t*>(s.data()), s.size());
  });
  m.def("tfrecord_read_all", &tfrecord_read_all, py::arg("path"), py::arg("verify") = true);
  // Region decode of JPEG files (jpeg_region.h).  `jpeg_header` -> (height, width, components) or
  // None; `jpeg_decode_region(bytes, y, x, h, w, denom)` -> (uint8 [rows, cols, 3], y0, x0,
  // image height, image width) in the coordinates of the image scaled by 1 / denom, or None when
  // the file needs the general decoder.
  m.def("jpeg_region_available", []() { return jpeg_region::available(); });
  m.def("jpeg_region_status", []() {
    const jpeg_region::Api& a = jpeg_region::api();
    return py::make_tuple(a.ok, a.version, (int)a.struct_size, a.why);
  });
  m.def("jpeg_header", [](py::bytes b) -> py::object {
    char* data; Py_ssize_t n;
    if (PyBytes_AsStringAndSize(b.ptr(), &data, &n)) throw py::error_already_set();
    int h = 0, w = 0, comps = 0;
    if (!jpeg_region::header(reinterpret_cast<const uint8_t*>(data), (size_t)n, &h, &w, &comps))
      return py::none();
    return py::make_tuple(h, w, comps);
  });
  m.def("jpeg_decode_region", [](py::bytes b, int y, int x, int h, int w, int denom) -> py::object {
    char* data; Py_ssize_t n;
    if (PyBytes_AsStringAndSize(b.ptr(), &data, &n)) throw py::error_already_set();
    jpeg_region::Region r;
    bool ok;
    {
      py::gil_scoped_release nogil;
      ok = jpeg_region::decode(reinterpret_cast<const uint8_t*>(data), (size_t)n, y, x, h, w, denom, &r);
    }
    if (!ok) return py::none();
    py::array_t<uint8_t> out({r.rows, r.cols, 3});
    std::memcpy(out.mutable_data(), r.pixels.data(), r.pixels.size());
    return py::make_tuple(out, r.y0, r.x0, r.image_h, r.image_w);
  }, py::arg("data"), py::arg("y"), py::arg("x"), py::arg("h"), py::arg("w"), py::arg("denom") = 1);
  m.def("crop_resize_bilinear", &crop_resize_bilinear, py::arg("image"), py::arg("y"), py::arg("x"),
        py::arg("h"), py::arg("w"), py::arg("resized_h"), py::arg("resized_w"), py::arg("top"),
        py::arg("left"), py::arg("out_h"), py::arg("out_w"), py::arg("flip") = false,
        py::arg("means") = std::vector<float>(), py::arg("origin_y") = 0, py::arg("origin_x") = 0);
  py::class_<CifarLoader>(m, "CifarLoader")
      .def(py::init<const std::vector<std::string>&, bool, int, int, int, uint64_t, int, bool, int>(),
           py::arg("files"), py::arg("is_training"), py::arg("batch_size"),
           py::arg("shard_index") = 0, py::arg("num_shards") = 1, py::arg("seed") = 0,
           py::arg("num_threads") = 4, py::arg("drop_remainder") = false,
           py::arg("num_epochs") = 0)
      .def("num_records", &CifarLoader::num_records)
      .def("next_batch", &CifarLoader::next_batch);
}
