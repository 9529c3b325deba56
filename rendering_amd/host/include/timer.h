// RAII millisecond timer printing "<name> <ms> ms" like the reference's include/timer.h.
#pragma once
#include <chrono>
#include <string>

class Timer {
public:
	explicit Timer(std::string name = "Unnamed timer:");
	~Timer();
	long long stop();
private:
	std::string name_;
	std::chrono::steady_clock::time_point start_;
	bool running_ = true;
};
