// image_io.cpp — see image_io.h
#include "image_io.h"

#include <fstream>
#include <iterator>
#include <stdexcept>

#include "png_reader.h"

namespace ngp {

void read_image_rgba8(const std::string& path, int& w, int& h, std::vector<uint8_t>& pixels) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error{"Could not open image file: " + path};
	const std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	if (bytes.size() >= 8 && bytes[0] == 0x89 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G') { decode_png_rgba8(bytes.data(), bytes.size(), w, h, pixels); return; }
	if (bytes.size() >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
		try { decode_jpeg_rgba8(bytes.data(), bytes.size(), w, h, pixels); }
		catch (const std::exception& e) { throw std::runtime_error{std::string{e.what()} + " (" + path + ")"}; }
		return;
	}
	if (is_hdr_signature(bytes.data(), bytes.size())) {
		try { decode_hdr_rgba8(bytes.data(), bytes.size(), w, h, pixels); }
		catch (const std::exception& e) { throw std::runtime_error{std::string{e.what()} + " (" + path + ")"}; }
		return;
	}
	throw std::runtime_error{"Could not open image file: unknown image type (PNG, JPEG and Radiance HDR are decoded; BMP / TGA / GIF / PSD / PIC / PNM are not): " + path};
}

} // namespace ngp
