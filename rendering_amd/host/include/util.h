// Helpers shared by the loaders (reference: include/util.h, src/util.cpp).
#pragma once
#include <string>
#include <string_view>
#include <vector>

#include "geometry.h"
#include "options.h"

#include <stdexcept>

// The reference's error behaviour (util.h:13-19): print "Error: <file> <func> <line>" and std::exit(-1).  Inside the C API
// used by the Python tests / bench (capi.cpp) the same condition is thrown as HostError instead and reported through
// rah_last_error(), so that a recoverable error (no GPU, missing asset, bad size) does not end the host process.
struct HostError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ThrowErrorsScope { ThrowErrorsScope(); ~ThrowErrorsScope(); bool prev; };
void noteError(const std::string& what);   // context printed before LOG_ERROR() is also kept for HostError::what()
#define LOG_ERROR() logError(__FILE__, __FUNCTION__, __LINE__)
[[noreturn]] void logError(const char* file, const char* func, int line);

inline float clamp(float lo, float hi, float v) { const float m = (v < hi) ? v : hi; return (lo < m) ? m : lo; }
inline float degToRad(float deg) { return deg * (float)(3.14159265358979323846) / 180.0f; }

bool strToBool(std::string_view s);
int strToInt(std::string_view s);
float strToFloat(std::string_view s);
Vec3f str3ToFloat(const std::vector<std::string>& parts);
std::vector<std::string> splitString(std::string_view s, char delim);

// BMP output with the reference's header bytes, bottom-up BGR rows and truncating quantiser
// (util.cpp:15-76; saturated channels are 255 -- SURVEY.md 0.4).  Does not launch an image viewer.
int saveImage(const Vec3f* frameBuffer, const Options& options);
int saveImageBGR(const unsigned char* bgrBottomUp, const Options& options);
// 24-bpp BMP, 54-byte header, no row padding; rows stay bottom-up, channels returned as RGB (util.cpp:78-113)
unsigned char* loadBMP(const char* filename, int& width, int& height);
