// Loader helpers, BMP I/O, timers and statistics of the host side (reference: src/util.cpp, include/util.h,
// timer.h, stats.h) -- own implementation.
#include "util.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <vector>

#include "stats.h"
#include "timer.h"

namespace {
thread_local bool gThrowErrors = false;
thread_local std::string gErrorNote;
}
ThrowErrorsScope::ThrowErrorsScope() : prev(gThrowErrors) { gThrowErrors = true; gErrorNote.clear(); }
ThrowErrorsScope::~ThrowErrorsScope() { gThrowErrors = prev; }
void noteError(const std::string& what) { gErrorNote = what; }

void logError(const char* file, const char* func, int line)
{
	// same message and exit status as the reference (util.h:13-19)
	std::cout << "Error: " << file << ' ' << func << ' ' << line << std::endl;
	if (gThrowErrors) throw HostError((gErrorNote.empty() ? std::string() : gErrorNote + " -- ") + "Error: " + file + " " + func + " " + std::to_string(line));
	std::exit(-1);
}

namespace {
template <typename T> T parseValue(std::string_view s)
{
	T v{};
	std::istringstream in{ std::string(s) };
	in >> v;
	if (!in.eof() && !in.good()) LOG_ERROR();
	return v;
}
}

bool strToBool(std::string_view s) { return parseValue<bool>(s); }
int strToInt(std::string_view s) { return parseValue<int>(s); }
float strToFloat(std::string_view s) { return parseValue<float>(s); }

Vec3f str3ToFloat(const std::vector<std::string>& p)
{
	if (p.size() != 3) LOG_ERROR();
	return Vec3f(strToFloat(p[0]), strToFloat(p[1]), strToFloat(p[2]));
}

std::vector<std::string> splitString(std::string_view s, char delim)
{
	std::vector<std::string> out;
	std::istringstream in{ std::string(s) };
	for (std::string cell; std::getline(in, cell, delim);) out.push_back(cell);
	return out;
}

namespace {
// The reference writes its 54-byte header with overlapping 8-byte stores (util.cpp:33-43); only the resulting
// bytes matter.  For W%4==0 they are a standard BITMAPINFOHEADER except biClrUsed/biClrImportant = 0.
void bmpHeader(unsigned char* h, size_t w, size_t ht)
{
	memset(h, 0, 54);
	const uint64_t arr = (uint64_t)ht * w * 3;
	auto put = [&](int off, uint64_t v) { memcpy(h + off, &v, 8); };
	h[0] = 'B'; h[1] = 'M';
	put(0x02, 54 + arr); put(0x0A, 54); put(0x0E, 40); put(0x12, w); put(0x16, ht);
	h[0x1A] = 1; h[0x1C] = 24;
	put(0x22, arr); put(0x26, 2835); put(0x2A, 2835);
}
}

int saveImageBGR(const unsigned char* bgr, const Options& options)
{
	if (options.width % 4 != 0) {
		std::cout << "saveImage: width must be a multiple of 4" << '\n';   // the reference corrupts memory here (util.cpp:28-57)
		return -2;
	}
	const std::string path = options.imageName + ".bmp";
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) { std::cout << "Could not open output file " << path << '\n'; return -1; }
	std::cout << "Successfully wrote to output file " << path << '\n';
	unsigned char hdr[54];
	bmpHeader(hdr, options.width, options.height);
	fwrite(hdr, 1, 54, f);
	fwrite(bgr, 1, options.width * options.height * 3, f);
	fclose(f);
	return 0;
}

int saveImage(const Vec3f* fb, const Options& options)
{
	std::vector<unsigned char> px(options.width * options.height * 3);
	unsigned char* p = px.data();
	for (size_t row = 0; row < options.height; ++row) {
		const Vec3f* src = fb + (options.height - 1 - row) * options.width;      // bottom-up
		for (size_t x = 0; x < options.width; ++x)
			for (int k = 2; k >= 0; --k) *p++ = (unsigned char)(int)(clamp(0.0f, 1.0f, src[x][k]) * 255);
	}
	return saveImageBGR(px.data(), options);
}

unsigned char* loadBMP(const char* filename, int& width, int& height)
{
	// Same result as the reference (util.cpp:78-113) for the files it can read -- 54-byte header, 24 bpp, no row
	// padding: rows stay bottom-up, channels come back as RGB.  Hardened: honours the pixel
	// data offset, row padding (width % 4 != 0), 32 bpp and top-down files (negative height) instead of
	// mis-reading them.
	FILE* f = fopen(filename, "rb");
	if (!f) {
		std::cout << "Could not open .bmp file: " << filename << '\n';
		LOG_ERROR();
	}
	unsigned char hdr[54];
	if (fread(hdr, 1, 54, f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') { fclose(f); std::cout << "Not a BMP file: " << filename << '\n'; LOG_ERROR(); }
	uint32_t dataOffset; int32_t w, h; uint16_t bpp;
	memcpy(&dataOffset, hdr + 10, 4); memcpy(&w, hdr + 18, 4); memcpy(&h, hdr + 22, 4); memcpy(&bpp, hdr + 28, 2);
	const bool topDown = h < 0;
	// the header is not trusted: dimensions are bounded (also rules out -INT_MIN), the pixel data must fit the file
	constexpr int32_t kMaxDim = 32768;
	if (w <= 0 || w > kMaxDim || h == 0 || h < -kMaxDim || h > kMaxDim || (bpp != 24 && bpp != 32)) {
		fclose(f); std::cout << "Unsupported BMP (need 24/32 bpp, 1..32768 pixels a side): " << filename << '\n'; noteError(std::string("unsupported BMP ") + filename); LOG_ERROR();
	}
	if (topDown) h = -h;
	width = w; height = h;
	if (dataOffset < 54) dataOffset = 54;
	const size_t bytesPP = bpp / 8, rowBytes = ((size_t)w * bytesPP + 3) & ~(size_t)3;
	long fileSize = -1;
	if (fseek(f, 0, SEEK_END) == 0) fileSize = ftell(f);
	if (fileSize < 0 || (size_t)fileSize < (size_t)dataOffset || fseek(f, (long)dataOffset, SEEK_SET) != 0) {
		fclose(f); std::cout << "Truncated BMP: " << filename << '\n'; noteError(std::string("truncated BMP ") + filename); LOG_ERROR();
	}
	std::vector<unsigned char> row(rowBytes);
	unsigned char* data = new unsigned char[(size_t)3 * (size_t)w * (size_t)h];
	for (int y = 0; y < h; ++y) {
		if (fread(row.data(), 1, rowBytes, f) != rowBytes) memset(row.data(), 0, rowBytes);   // truncated file: black
		unsigned char* dst = data + (size_t)3 * (size_t)w * (size_t)(topDown ? (h - 1 - y) : y);             // keep file order = bottom-up
		for (int x = 0; x < w; ++x) {
			dst[(size_t)x * 3 + 0] = row[(size_t)x * bytesPP + 2];     // BGR -> RGB
			dst[(size_t)x * 3 + 1] = row[(size_t)x * bytesPP + 1];
			dst[(size_t)x * 3 + 2] = row[(size_t)x * bytesPP + 0];
		}
	}
	fclose(f);
	return data;
}

// ---- options / stats / timer -------------------------------------------------------------------
void options::reset()
{
	outputProgress = true; useBackfaceCulling = true; collectStatistics = false; enableOutput = true;
	imageOutput = true; useAC = true; showAC = false; useSkybox = false; useTextures = true;
	showNormals = false; enableSSAA = true;
}

void stats::reset() { rayTriTests = accelStructTests = triCopiesCount = meshCount = acCount = raysCasted = 0; }

void stats::printStats()
{
	std::cout.precision(2);
	std::cout << "Statistics:\n";
	auto line = [](const char* name, double v) { std::cout << std::left << std::setw(36) << name << std::setw(10) << std::scientific << v << '\n'; };
	line("Ray triangle tests:", (double)rayTriTests);
	line("Ray acceleration structure tests:", (double)accelStructTests);
	line("Total intersection test:", (double)rayTriTests + (double)accelStructTests);
	std::cout << std::left << std::setw(36) << "Total triangle copies:" << triCopiesCount << '\n';
	std::cout << std::left << std::setw(36) << "Total triangle count:" << meshCount << '\n';
	std::cout << std::left << std::setw(36) << "Acceleration structure count:" << acCount << '\n';
	line("Rays casted:", (double)raysCasted);
}

Timer::Timer(std::string name) : name_(std::move(name)), start_(std::chrono::steady_clock::now()) {}
Timer::~Timer() { stop(); }
long long Timer::stop()
{
	if (!running_) return 0;
	running_ = false;
	const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start_).count();
	if (options::enableOutput) std::cout << std::setw(18) << std::left << name_ << ms << " ms" << std::endl;
	return ms;
}
