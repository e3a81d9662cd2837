// oracle/ref_imageio.cpp — TEST INFRASTRUCTURE ONLY.  Thin C entry points over the reference's OWN image decoders: the vendored stb_image and
// tinyexr headers are compiled where they lie under /root/reference/dependencies (nothing is copied into this repository); the result,
// oracle/_ref/libref_imageio.so, is what tests/test_image_io_cpu.py holds this build's PNG / JPEG / EXR decoders (blender-ngp_amd/host/*_reader.cpp)
// against.  The calls are the ones the reference's loader makes: stbi_load(path, &w, &h, &comp, 4) (src/nerf_loader.cu:581, 589, 606),
// stbi_load_16(path, &w, &h, &comp, 1) (636), LoadEXR (src/tinyexr_wrapper.cu:120-134).
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image/stb_image.h>
#define TINYEXR_IMPLEMENTATION
#include <tinyexr/tinyexr.h>
#include <vector>

extern "C" {
unsigned char* ref_stbi_load_rgba8(const char* path, int* w, int* h) { int comp = 0; return stbi_load(path, w, h, &comp, 4); }
unsigned short* ref_stbi_load_16_gray(const char* path, int* w, int* h) { int comp = 0; return stbi_load_16(path, w, h, &comp, 1); }
float* ref_load_exr_rgba(const char* path, int* w, int* h) { float* data = nullptr; const char* err = nullptr; if (LoadEXR(&data, w, h, path, &err) != TINYEXR_SUCCESS) { if (err) FreeEXRErrorMessage(err); return nullptr; } return data; }
// fixture generator (tests/golden/make_exr_fixtures.py): tinyexr's own writer, scanline files, compression = TINYEXR_COMPRESSIONTYPE_*, pixel_type HALF 1 / FLOAT 2
int ref_save_exr_rgba(const char* path, const float* rgba, int w, int h, int compression, int pixel_type) {
	EXRHeader header; InitEXRHeader(&header);
	EXRImage image; InitEXRImage(&image);
	image.num_channels = 4;
	std::vector<float> planes[4];
	for (int c = 0; c < 4; ++c) { planes[c].resize((size_t)w * h); for (size_t i = 0; i < (size_t)w * h; ++i) planes[c][i] = rgba[4 * i + c]; }
	float* ptrs[4] = {planes[3].data(), planes[2].data(), planes[1].data(), planes[0].data()};   // A, B, G, R: the alphabetical order EXR stores
	image.images = (unsigned char**)ptrs; image.width = w; image.height = h;
	header.num_channels = 4;
	header.channels = (EXRChannelInfo*)malloc(sizeof(EXRChannelInfo) * 4);
	const char* names = "ABGR";
	for (int c = 0; c < 4; ++c) { header.channels[c].name[0] = names[c]; header.channels[c].name[1] = 0; }
	header.pixel_types = (int*)malloc(sizeof(int) * 4); header.requested_pixel_types = (int*)malloc(sizeof(int) * 4);
	for (int c = 0; c < 4; ++c) { header.pixel_types[c] = TINYEXR_PIXELTYPE_FLOAT; header.requested_pixel_types[c] = pixel_type; }
	header.compression_type = compression;
	const char* err = nullptr;
	const int r = SaveEXRImageToFile(&image, &header, path, &err);
	if (err) FreeEXRErrorMessage(err);
	free(header.channels); free(header.pixel_types); free(header.requested_pixel_types);
	return r;
}
const char* ref_stbi_failure_reason() { return stbi_failure_reason(); }
void ref_free(void* p) { free(p); }
}
