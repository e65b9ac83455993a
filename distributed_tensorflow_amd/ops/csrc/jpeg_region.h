// Region decoding of JPEG files for the ImageNet training pipeline: only the rows and (MCU-aligned)
// columns of the sampled crop box go through the inverse DCT, upsampling and colour conversion,
// optionally at 1/2, 1/4 or 1/8 scale.  This is what the reference gets from
// `tf.image.decode_and_crop_jpeg` (`image_classification/imagenet_preprocessing.py:259-323`).
//
// The decoder is the system's libjpeg-turbo (`libjpeg.so.8` / `libjpeg.so.62`), loaded with dlopen:
// its partial-decode entry points `jpeg_crop_scanline` / `jpeg_skip_scanlines` exist since 1.5.
// The image ships the library without headers, so the few declarations needed are written out
// here.  Only the leading fields of `jpeg_decompress_struct` are touched - their layout has not
// changed since libjpeg 6b (the offsets are pinned by static_asserts against the values of a
// reference header) - and the object itself is an oversized buffer whose official size and API
// version are asked from the library at start-up: `jpeg_CreateDecompress` reports both in the
// error it raises for a mismatch.  Everything fails soft: `available()` is false when the library
// or one of the entry points is missing, and callers fall back to PIL.
#pragma once
#include <dlfcn.h>
#include <setjmp.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace jpeg_region {

struct ErrorMgr {                       // struct jpeg_error_mgr
  void (*error_exit)(void*);
  void (*emit_message)(void*, int);
  void (*output_message)(void*);
  void (*format_message)(void*, char*);
  void (*reset_error_mgr)(void*);
  int msg_code;
  union { int i[8]; char s[80]; } msg_parm;
  int trace_level;
  long num_warnings;
  const char* const* jpeg_message_table;
  int last_jpeg_message;
  const char* const* addon_message_table;
  int first_addon_message;
  int last_addon_message;
};
static_assert(offsetof(ErrorMgr, msg_code) == 40 && offsetof(ErrorMgr, msg_parm) == 44 &&
              offsetof(ErrorMgr, trace_level) == 124 && offsetof(ErrorMgr, num_warnings) == 128 &&
              sizeof(ErrorMgr) == 168, "jpeg_error_mgr layout");

struct DecompressHead {                 // leading fields of struct jpeg_decompress_struct
  ErrorMgr* err;
  void* mem;
  void* progress;
  void* client_data;
  int is_decompressor;
  int global_state;
  void* src;
  unsigned image_width, image_height;
  int num_components;
  int jpeg_color_space;
  int out_color_space;
  unsigned scale_num, scale_denom;
  double output_gamma;
  int buffered_image, raw_data_out;
  int dct_method;
  int do_fancy_upsampling, do_block_smoothing, quantize_colors;
  int dither_mode;
  int two_pass_quantize, desired_number_of_colors;
  int enable_1pass_quant, enable_external_quant, enable_2pass_quant;
  unsigned output_width, output_height;
  int out_color_components, output_components;
  int rec_outbuf_height;
  int actual_number_of_colors;
  void* colormap;
  unsigned output_scanline;
};
static_assert(offsetof(DecompressHead, src) == 40 && offsetof(DecompressHead, image_width) == 48 &&
              offsetof(DecompressHead, num_components) == 56 &&
              offsetof(DecompressHead, jpeg_color_space) == 60 &&
              offsetof(DecompressHead, out_color_space) == 64 &&
              offsetof(DecompressHead, scale_num) == 68 && offsetof(DecompressHead, output_gamma) == 80 &&
              offsetof(DecompressHead, dct_method) == 96 && offsetof(DecompressHead, output_width) == 136 &&
              offsetof(DecompressHead, output_components) == 148 &&
              offsetof(DecompressHead, output_scanline) == 168, "jpeg_decompress_struct layout");

constexpr int kJcsGrayscale = 1, kJcsRgb = 2, kJcsYCbCr = 3;
constexpr size_t kObjectBytes = 4096;   // the real struct has 600 - 700 bytes

struct Api {
  ErrorMgr* (*std_error)(ErrorMgr*) = nullptr;
  void (*create)(void*, int, size_t) = nullptr;
  void (*destroy)(void*) = nullptr;
  void (*mem_src)(void*, const unsigned char*, unsigned long) = nullptr;
  int (*read_header)(void*, int) = nullptr;
  int (*start)(void*) = nullptr;
  unsigned (*read_scanlines)(void*, unsigned char**, unsigned) = nullptr;
  unsigned (*skip_scanlines)(void*, unsigned) = nullptr;
  void (*crop_scanline)(void*, unsigned*, unsigned*) = nullptr;
  int version = 0;
  size_t struct_size = 0;
  bool ok = false;
  std::string why;
};

struct Jump {
  ErrorMgr mgr;                         // first member: cinfo->err points here
  jmp_buf env;
};

inline void on_error(void* cinfo) {
  Jump* j = reinterpret_cast<Jump*>(reinterpret_cast<DecompressHead*>(cinfo)->err);
  longjmp(j->env, 1);
}
inline void on_message(void*, int) {}   // warnings about corrupt data: keep decoding, like PIL

inline const Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    for (const char* name : {"libjpeg.so.8", "libjpeg.so.62", "libjpeg.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib && dlsym(lib, "jpeg_crop_scanline") && dlsym(lib, "jpeg_skip_scanlines")) break;
      lib = nullptr;
    }
    if (!lib) { a.why = "no libjpeg-turbo with jpeg_crop_scanline / jpeg_skip_scanlines found"; return; }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    a.std_error = reinterpret_cast<decltype(a.std_error)>(sym("jpeg_std_error"));
    a.create = reinterpret_cast<decltype(a.create)>(sym("jpeg_CreateDecompress"));
    a.destroy = reinterpret_cast<decltype(a.destroy)>(sym("jpeg_destroy_decompress"));
    a.mem_src = reinterpret_cast<decltype(a.mem_src)>(sym("jpeg_mem_src"));
    a.read_header = reinterpret_cast<decltype(a.read_header)>(sym("jpeg_read_header"));
    a.start = reinterpret_cast<decltype(a.start)>(sym("jpeg_start_decompress"));
    a.read_scanlines = reinterpret_cast<decltype(a.read_scanlines)>(sym("jpeg_read_scanlines"));
    a.skip_scanlines = reinterpret_cast<decltype(a.skip_scanlines)>(sym("jpeg_skip_scanlines"));
    a.crop_scanline = reinterpret_cast<decltype(a.crop_scanline)>(sym("jpeg_crop_scanline"));
    if (!a.std_error || !a.create || !a.destroy || !a.mem_src || !a.read_header || !a.start ||
        !a.read_scanlines) { a.why = "libjpeg entry point missing"; return; }
    // Ask the library for its API version and the size of its decompress object: a call with the
    // wrong value raises an error whose first parameter is the value the library wants.
    std::vector<uint64_t> obj(kObjectBytes / 8, 0);
    Jump j;
    auto probe = [&](int version, size_t size, int* wanted) {
      std::memset(obj.data(), 0, kObjectBytes);
      a.std_error(&j.mgr);
      j.mgr.error_exit = on_error;
      j.mgr.emit_message = on_message;
      reinterpret_cast<DecompressHead*>(obj.data())->err = &j.mgr;
      if (setjmp(j.env)) { *wanted = j.mgr.msg_parm.i[0]; return false; }
      a.create(obj.data(), version, size);
      a.destroy(obj.data());
      return true;
    };
    int wanted = 0;
    if (probe(0, 0, &wanted)) { a.why = "libjpeg accepted API version 0"; return; }
    a.version = wanted;
    if (a.version < 62 || a.version > 99) { a.why = "unexpected libjpeg API version"; return; }
    if (probe(a.version, 0, &wanted)) { a.why = "libjpeg accepted an empty object"; return; }
    a.struct_size = (size_t)wanted;
    if (a.struct_size < sizeof(DecompressHead) || a.struct_size > kObjectBytes / 2) {
      a.why = "unexpected size of jpeg_decompress_struct";
      return;
    }
    if (!probe(a.version, a.struct_size, &wanted)) { a.why = "libjpeg rejected its own object size"; return; }
    a.ok = true;
  });
  return a;
}

inline bool available() { return api().ok; }

struct Region {
  std::vector<uint8_t> pixels;          // [rows][cols][3] RGB
  int rows = 0, cols = 0;
  int y0 = 0, x0 = 0;                   // position of pixels[0][0] in the (scaled) image
  int image_h = 0, image_w = 0;         // size of the (scaled) image
};

// Dimensions and component count from the frame header (no pixel work).  False: not a JPEG the
// region decoder handles (the caller uses its general decoder).
inline bool header(const uint8_t* data, size_t n, int* h, int* w, int* comps) {
  const Api& a = api();
  if (!a.ok || n < 4 || data[0] != 0xff || data[1] != 0xd8) return false;
  std::vector<uint64_t> obj(kObjectBytes / 8, 0);
  DecompressHead* c = reinterpret_cast<DecompressHead*>(obj.data());
  Jump j;
  a.std_error(&j.mgr);
  j.mgr.error_exit = on_error;
  j.mgr.emit_message = on_message;
  c->err = &j.mgr;
  volatile bool created = false;
  if (setjmp(j.env)) {
    if (created) a.destroy(c);
    return false;
  }
  a.create(c, a.version, a.struct_size);
  created = true;
  a.mem_src(c, data, (unsigned long)n);
  a.read_header(c, 1);
  *h = (int)c->image_height;
  *w = (int)c->image_width;
  *comps = c->num_components;
  a.destroy(c);
  return true;
}

// Decodes rows [y, y + h) and at least columns [x, x + w) of the image scaled by 1 / denom
// (denom 1, 2, 4 or 8; the box is given in the scaled image's coordinates and is clamped to it).
// False: the file needs the general decoder (CMYK / YCCK, not a JPEG, damaged beyond use).
inline bool decode(const uint8_t* data, size_t n, int y, int x, int h, int w, int denom, Region* out) {
  const Api& a = api();
  if (!a.ok || n < 4 || data[0] != 0xff || data[1] != 0xd8) return false;
  if (denom != 1 && denom != 2 && denom != 4 && denom != 8) return false;
  std::vector<uint64_t> obj(kObjectBytes / 8, 0);
  DecompressHead* c = reinterpret_cast<DecompressHead*>(obj.data());
  Jump j;
  a.std_error(&j.mgr);
  j.mgr.error_exit = on_error;
  j.mgr.emit_message = on_message;
  c->err = &j.mgr;
  volatile bool created = false;
  if (setjmp(j.env)) {
    if (created) a.destroy(c);
    return false;
  }
  a.create(c, a.version, a.struct_size);
  created = true;
  a.mem_src(c, data, (unsigned long)n);
  a.read_header(c, 1);
  if (c->jpeg_color_space != kJcsGrayscale && c->jpeg_color_space != kJcsYCbCr &&
      c->jpeg_color_space != kJcsRgb) {
    a.destroy(c);
    return false;
  }
  c->out_color_space = kJcsRgb;
  c->scale_num = 1;
  c->scale_denom = (unsigned)denom;
  a.start(c);
  const int H = (int)c->output_height, W = (int)c->output_width;
  if (c->output_components != 3 || H <= 0 || W <= 0) { a.destroy(c); return false; }
  y = y < 0 ? 0 : (y >= H ? H - 1 : y);
  x = x < 0 ? 0 : (x >= W ? W - 1 : x);
  h = h < 1 ? 1 : (y + h > H ? H - y : h);
  w = w < 1 ? 1 : (x + w > W ? W - x : w);
  unsigned xoff = (unsigned)x, width = (unsigned)w;
  if (x != 0 || w != W) a.crop_scanline(c, &xoff, &width);   // moves xoff down to an MCU boundary
  out->rows = h;
  out->cols = (int)width;
  out->y0 = y;
  out->x0 = (int)xoff;
  out->image_h = H;
  out->image_w = W;
  out->pixels.resize((size_t)h * width * 3);
  if (y > 0 && a.skip_scanlines(c, (unsigned)y) != (unsigned)y) { a.destroy(c); return false; }
  int got = 0;
  while (got < h) {
    unsigned char* row = out->pixels.data() + (size_t)got * width * 3;
    const unsigned r = a.read_scanlines(c, &row, 1);
    if (r == 0) { a.destroy(c); return false; }    // truncated file
    got += (int)r;
  }
  a.destroy(c);                          // rows below the box are never decoded
  return true;
}

}  // namespace jpeg_region
