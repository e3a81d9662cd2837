// oracle/ref_imageio.cpp — TEST INFRASTRUCTURE ONLY.  Thin C entry points over the reference's OWN image decoders: the vendored stb_image and
// tinyexr headers are compiled where they lie under /root/reference/dependencies (nothing is copied into this repository); the result,
// oracle/_ref/libref_imageio.so, is what tests/test_image_io_cpu.py holds this build's PNG / JPEG / EXR decoders (blender-ngp_amd/host/*_reader.cpp)
// against.  The calls are the ones the reference's loader makes: stbi_load(path, &w, &h, &comp, 4) (src/nerf_loader.cu:581, 589, 606),
// stbi_load_16(path, &w, &h, &comp, 1) (636), LoadEXR (src/tinyexr_wrapper.cu:120-134).
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image/stb_image.h>
#define TINYEXR_IMPLEMENTATION
#include <tinyexr/tinyexr.h>

extern "C" {
unsigned char* ref_stbi_load_rgba8(const char* path, int* w, int* h) { int comp = 0; return stbi_load(path, w, h, &comp, 4); }
unsigned short* ref_stbi_load_16_gray(const char* path, int* w, int* h) { int comp = 0; return stbi_load_16(path, w, h, &comp, 1); }
float* ref_load_exr_rgba(const char* path, int* w, int* h) { float* data = nullptr; const char* err = nullptr; if (LoadEXR(&data, w, h, path, &err) != TINYEXR_SUCCESS) { if (err) FreeEXRErrorMessage(err); return nullptr; } return data; }
const char* ref_stbi_failure_reason() { return stbi_failure_reason(); }
void ref_free(void* p) { free(p); }
}
